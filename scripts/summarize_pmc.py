#!/usr/bin/env python3
"""Markdown summary of three rocprofv3 --pmc passes of `bench.py --no-graph --in-flight 1` (SQ counters; FETCH_SIZE +
GRBM_GUI_ACTIVE; WRITE_SIZE).  Usage: summarize_pmc.py <round label> <pmc_sq dir> <pmc_fetch dir> <pmc_write dir>"""
import argparse
import collections
import csv
import glob
import json
import re
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:60]


_HAS_STEM = [None]


def opens_forward(k):
    """the first kernel of every forward: the one-launch stem, or the layout kernel where the stem is three launches"""
    return ('stem_kernel' in k) if _HAS_STEM[0] else ('nchw_to_nhwc' in k)


def is_gemm(k):
    """the matrix launches of a forward: both conv kernel families, the one-launch bottleneck and the one-launch stem"""
    return 'conv_igemm' in k or 'conv_dma' in k or 'bottleneck_kernel' in k or 'stem_kernel' in k


def template_args(k):
    """the template arguments of a kernel name as strings ([] without a template list)"""
    m = re.search(r'<([^>]*)>', k)
    return m.group(1).replace(' ', '').split(',') if m else []


def arith(k):
    """The arithmetic of a matrix launch from its kernel name, read BY POSITION (the templates grow at the tail):
    conv_igemm_kernel<BM, BN, WM, WN, MINW, NBUF, BK, PREC, ...>: argument 7 (absent = the default = f32): 0 f32, 1 bf16x3, 2 fp16x2;
    conv_dma_kernel<BM, BK, WAVES_K, S, BALANCED, WAVES_N, NPL, CHAN, ...>: argument 6 (absent = the default = 3): 3 bf16x3, 2 fp16x2."""
    args = template_args(k)
    if 'stem_kernel' in k:
        return 'fp16x2 (one-launch stem)'
    if 'bottleneck_kernel' in k:
        return 'fp16x2 (one-launch bottleneck)'
    if 'conv_dma' in k:
        npl = args[6] if len(args) > 6 else '3'
        if npl not in ('2', '3'):
            raise ValueError(f"conv_dma_kernel: unknown pieces-per-operand argument {npl!r} in {k!r}")
        return {'2': 'fp16x2 (LDS-DMA)', '3': 'bf16x3 (LDS-DMA)'}[npl]
    if 'conv_igemm' in k:
        prec = args[7] if len(args) > 7 else '0'
        if prec not in ('0', '1', '2'):
            raise ValueError(f"conv_igemm_kernel: unknown arithmetic argument {prec!r} in {k!r}")
        return {'0': 'f32', '1': 'bf16x3', '2': 'fp16x2'}[prec]
    return 'f32'


def load(d):
    cc = sorted(glob.glob(d + '/*/*counter_collection.csv'))[-1]
    kt = sorted(glob.glob(d + '/*/*kernel_trace.csv'))[-1]
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt))}
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(cc)):
        per[short(r['Kernel_Name'])][r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    return per, dur


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("label"); ap.add_argument("sq"); ap.add_argument("fetch"); ap.add_argument("write")
    ap.add_argument("--layers", default=None, help="bench.py --dump-layers JSON: adds the per-layer table of the last forward")
    ap.add_argument("--tcc", default=None, help="a fourth pass with TCC_HIT_sum TCC_MISS_sum (L2 hit rate per layer)")
    ap.add_argument("--workload", default="B=1, 3x800x1333, ResNet-50")
    ap.add_argument("--batch", type=int, default=1, help="images per forward of the profiled workload: scales the algorithmic-MB column")
    ap.add_argument("--backbone", default="resnet50")
    A = ap.parse_args(argv)
    label = A.label
    sq, dsq = load(A.sq); fe, dfe = load(A.fetch); wr, dwr = load(A.write)
    _HAS_STEM[0] = any('stem_kernel' in k for k in sq)
    print(f"# {label} - PMC counters, `bench.py --no-graph --in-flight 1` ({A.workload}), rocprofv3 --pmc, separate passes\n")
    print("Per dispatch, averaged over the dispatches of the second half of the forwards (the timed steps; the first half contains plan")
    print("building and warm-up; kernels that only ran before that - tuning candidates - are not listed).  `FETCH_SIZE` / `WRITE_SIZE` are KiB; on gfx950 `FETCH_SIZE` under-reports wide")
    print("coalesced reads by 2x (MI355X_MICROARCH.md, HBM section), so HBM read bytes = 2 x FETCH_SIZE.  Commands: profiles/README.md.\n")
    g = [c.get('GRBM_GUI_ACTIVE') for k in fe if is_gemm(k) for _, c in fe[k].items() if c.get('GRBM_GUI_ACTIVE')]
    d = [dfe[i] for k in fe if is_gemm(k) for i, c in fe[k].items() if c.get('GRBM_GUI_ACTIVE')]
    clock_meas = sum(g) / 8 / (sum(d) * 1e-6) if d else 2.4e9
    clock = min(clock_meas, 2.4e9)
    print("## Matrix-core utilisation of the conv GEMMs (pass 1: SQ counters)\n")
    print("`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x 2.4 GHz).  The template's last argument is the")
    print("arithmetic.  `conv_igemm_kernel<..., PREC>`: 0 / absent = f32 MFMA (v_mfma_f32_32x32x2_f32), 1 = bf16x3 (v_mfma_f32_32x32x16_bf16,")
    print("six per f32 product), 2 = fp16x2 (v_mfma_f32_32x32x16_f16, three per f32 product); `conv_dma_kernel<..., NPL>`: 3 = bf16x3, 2 = fp16x2.\n")
    print("| kernel | arithmetic | dispatches | avg us | MFMA busy | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE |")
    print("|---|---|---:|---:|---:|---:|---:|")
    tot = collections.defaultdict(lambda: [0.0, 0.0])
    def forward_starts(per):
        """dispatch ids of the layout kernel that opens every forward (eager bench: one per step)"""
        return sorted(int(i) for k in per if opens_forward(k) for i in per[k])


    def timed_only(per, items):
        """keep the dispatches of the LAST HALF of the forwards: no plan building, no tuning candidates (with a tiles file that
        carries the head choices there are none anyway)"""
        st = forward_starts(per)
        if not st:
            return items[len(items) // 2:]
        lo = st[len(st) // 2]
        return [(i, c) for i, c in items if int(i) >= lo]


    for k in sorted(sq):
        if not is_gemm(k):
            continue
        ds = timed_only(sq, list(sq[k].items()))
        n = len(ds)
        if n == 0:
            continue
        us = sum(dsq[i] for i, _ in ds) / n
        avg = lambda name: sum(c.get(name, 0) for _, c in ds) / n
        mf, wc, wi, bc, la = avg('SQ_VALU_MFMA_BUSY_CYCLES'), avg('SQ_WAVE_CYCLES'), avg('SQ_WAIT_INST_ANY'), avg('SQ_LDS_BANK_CONFLICT'), avg('SQ_LDS_IDX_ACTIVE')
        cyc = 4 * 256 * us * 1e-6 * clock
        a = arith(k)
        tot[a][0] += mf * n; tot[a][1] += cyc * n
        print(f"| `{k}` | {a} | {n} | {us:.1f} | {100 * mf / cyc:.1f}% | {wi / max(wc, 1):.2f} | {bc / max(la, 1):.3f} |")
    for a, (m, c) in tot.items():
        print(f"\nAll {a} conv dispatches together: MFMA pipes busy **{100 * m / c:.1f}%** of the time these kernels run.")
    print("\n## HBM traffic of the memory-bound kernels (passes 2 and 3: FETCH_SIZE, WRITE_SIZE)\n")
    print("| kernel | avg us | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM MB = (2*FETCH+WRITE)*1024/1e6 | achieved GB/s (PMC bytes / time) | algorithmic MB (DESIGN.md section 4) |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    # per IMAGE at 3x800x1333; the column is scaled by --batch.  RoI pooling: feature map + K*C*4 out, by backbone
    alg1 = {'maxpool3x3s2_kernel': 85.4, 'nchw_to_nhwc_small_kernel': 29.9,
            'rpn_decode_kernel': 0.42 if A.backbone == 'resnet50' else 1.66,
            'roi_pool_avg_kernel': 11.1 if A.backbone == 'resnet50' else 9.2,
            'nms_mask_kernel<4, -1>': 1.18, 'nms_scan_kernel<1>': 1.13, 'nms_scan_kernel': 1.13, 'sort_topk_kernel<10>': 0.11, 'detections_kernel': 0.5}
    alg = {k: round(v * A.batch, 2) for k, v in alg1.items()}
    mem_kernels = ['nchw_to_nhwc_small_kernel', 'maxpool3x3s2_kernel', 'rpn_decode_kernel', 'sort_topk_kernel<10>',
                   'nms_mask_kernel<4, -1>', 'nms_scan_kernel<1>', 'nms_scan_kernel', 'roi_pool_avg_kernel', 'detections_kernel']
    # HarDNet / ResNeXt: the depthwise, pair and grouped kernels (HBM-bound; algorithmic bytes differ per layer: not tabulated, the
    # achieved GB/s column is the figure of merit) and the top-k by rank
    mem_kernels += sorted(k for k in fe if any(t in k for t in ('dwconv3x3_kernel', 'gconv1x1_pair_kernel', 'gconv3x3_kernel', 'topk_rank_kernel', 'absmax_kernel')))
    for k in mem_kernels:
        if k not in fe:
            continue
        ds = list(fe[k].items()); ds = ds[len(ds) // 2:]
        n = len(ds); us = sum(dfe[i] for i, _ in ds) / n
        f = sum(c.get('FETCH_SIZE', 0) for _, c in ds) / n
        dw = list(wr[k].items()); dw = dw[len(dw) // 2:]
        w = sum(c.get('WRITE_SIZE', 0) for _, c in dw) / max(1, len(dw))
        mb = (2 * f + w) * 1024 / 1e6
        print(f"| `{k}` | {us:.1f} | {f:.0f} | {w:.0f} | {mb:.2f} | {mb * 1e6 / (us * 1e-6) / 1e9:.0f} | {alg.get(k, '')} |")
    print(f"\nClock check: sum(GRBM_GUI_ACTIVE)/8/sum(duration) over the conv dispatches = {clock_meas / 1e9:.2f} GHz (reads high on dispatches")
    print("shorter than ~0.3 ms; the table uses min(measured, 2.4 GHz)).")


    def last_forward(per):
        start = max(int(i) for k in per if opens_forward(k) for i in per[k])
        out = collections.defaultdict(lambda: [0, 0.0])
        for k in per:
            fam = 'conv_igemm_kernel + conv_dma_kernel' if is_gemm(k) else k
            for i, c in per[k].items():
                if int(i) >= start:
                    out[fam][0] += 1
                    out[fam][1] += sum(v for n, v in c.items() if n in ('FETCH_SIZE', 'WRITE_SIZE'))
        return out


    lf, lw = last_forward(fe), last_forward(wr)
    print("\n## HBM traffic of the conv GEMMs in one forward (last forward of passes 2 and 3)\n")
    print("| kernel family | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM MB = (2*FETCH+WRITE)*1024/1e6 |")
    print("|---|---:|---:|---:|---:|")
    fam = 'conv_igemm_kernel + conv_dma_kernel'
    if fam in lf:
        mb = (2 * lf[fam][1] + lw[fam][1]) * 1024 / 1e6
        print(f"| `{fam}` | {lf[fam][0]} | {lf[fam][1]:.0f} | {lw[fam][1]:.0f} | {mb:.1f} |")
        print(f"\n{mb / 1e3:.2f} GB per forward over all {lf[fam][0]} matrix launches (the trunk's convs and one-launch bottlenecks + fused RPN conv + fused head GEMM; K-slice slabs")
        print("are written and read inside these launches now, there is no reduce kernel).  The excess over the algorithmic bytes is")
        print("(a) the K-slice partial slabs (write-through stores, read back by the tile's last-arriving slice), (b) the activation")
        print("tile re-read by every output-channel tile of its row block once it has left the XCD's L2, (c) the 7x8x4 stem reading its")
        print("4-channel input 7 times, (d) for bf16x3 layers the pre-split weight image (6 instead of 4 bytes per weight).")


    def last_forward_convs(per, dur):
        """[(dispatch id, kernel, counters, us)] of the conv GEMMs of the last forward, in launch order"""
        st = forward_starts(per)
        lo = st[-1]
        rows = [(int(i), k, c, dur[i]) for k in per if is_gemm(k) for i, c in per[k].items() if int(i) >= lo]
        return sorted(rows)


    if A.layers:
        layers = json.load(open(A.layers))
        sqr, fer, wrr = last_forward_convs(sq, dsq), last_forward_convs(fe, dfe), last_forward_convs(wr, dwr)
        tcr = None
        if A.tcc:
            tc, dtc = load(A.tcc)
            tcr = last_forward_convs(tc, dtc)
        print("\n## Per conv layer (last forward of every pass, launch order; names, algorithmic and slab bytes from `bench.py --dump-layers`)\n")
        print("`HBM MB` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / 1e6 of that dispatch; `slab MB` = the K-slice partial slabs of the chosen")
        print("schedule, counted twice (written write-through by the slices, read back by the last arriver): the part of the excess that is the")
        print("price of filling the chip by cutting K; `x alg` = HBM MB / algorithmic MB; `L2 hit` = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum).\n")
        print("| # | layer | tile | arithmetic | split | us | MFMA busy | HBM MB | algorithmic MB | x alg | slab MB (w+r) | HBM - slab, x alg | L2 hit |")
        print("|---:|---|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
        tot = [0.0, 0.0, 0.0, 0.0]
        for j, L in enumerate(layers):
            if j >= len(sqr) or j >= len(fer) or j >= len(wrr):
                break
            us = sqr[j][3]
            mf = sqr[j][2].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * 256 * us * 1e-6 * clock)
            mb = (2 * fer[j][2].get('FETCH_SIZE', 0) + wrr[j][2].get('WRITE_SIZE', 0)) * 1024 / 1e6
            alg_mb, slab = L['algorithmic_bytes'] / 1e6, 2 * L['slab_bytes'] / 1e6
            # the workspace is sized for every slab the schedule COULD use; balanced / hybrid schedules touch all of them
            hit = ''
            if tcr is not None and j < len(tcr):
                h, m = tcr[j][2].get('TCC_HIT_sum', 0), tcr[j][2].get('TCC_MISS_sum', 0)
                hit = f"{100 * h / max(1, h + m):.0f}%"
            tot[0] += us; tot[1] += mb; tot[2] += alg_mb; tot[3] += slab
            print(f"| {j} | {L['name']} | {L['tile']} | {('f32', 'bf16x3', 'fp16x2')[int(L.get('precision', 0))]} | {L['split_k']} | {us:.1f} | {100 * mf:.1f}% | {mb:.1f} | {alg_mb:.1f} | {mb / alg_mb:.2f} | "
                  f"{slab:.1f} | {(mb - slab) / alg_mb:.2f} | {hit} |")
        print(f"| | **all {len(layers)} trunk convs** | | | | {tot[0]:.0f} | | {tot[1]:.0f} | {tot[2]:.0f} | {tot[1] / tot[2]:.2f} | {tot[3]:.0f} | "
              f"{(tot[1] - tot[3]) / tot[2]:.2f} | |")


if __name__ == "__main__":
    main()
