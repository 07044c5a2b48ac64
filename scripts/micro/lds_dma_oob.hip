// Semantics probe: what does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` leave in LDS -- zeros or the old bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i make_rsrc(const void *p, unsigned bytes) {
    v4i r; unsigned long long a = (unsigned long long)p;
    r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffff); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}
__global__ void k(const float *src, unsigned bytes, float *out) {
    __shared__ __align__(16) float lds[64 * 4 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = -7.f;
    __syncthreads();
    v4i rsrc = make_rsrc(src, bytes);
    // lanes 0..31 in range (reversed order: lane l reads 16 B at (31-l)*16), lanes 32..47 out of range by offset -16, 48..63 past the end
    unsigned voff = lane < 32 ? (31 - lane) * 16 : (lane < 48 ? 0xfffffff0u : bytes + (lane - 48) * 16);
    unsigned ldsbase = (unsigned)(size_t)(lds) ;   // LDS byte address (low 32 bits of the generic->local address)
    ldsbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(rsrc), "s"(ldsbase) : "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    float h[128]; for (int i = 0; i < 128; ++i) h[i] = (float)i;
    float *src, *out; hipMalloc(&src, sizeof(h)); hipMalloc(&out, 512 * 4);
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(src, 512, out);
    float o[512]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d: %6.1f %6.1f %6.1f %6.1f\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]); }
    printf("after: %6.1f\n", o[256]);
    return 0;
}
