import sys, ctypes, torch
sys.path.insert(0, '/root/repo')
from two_stage_object_detection_amd import _ffi
torch.zeros(1, device='cuda')
L = _ffi.lib()
for t in (1, 2, 3, 4):
    print('tile', _ffi.TILE_NAMES[t], 'occupancy blocks/CU', L.tsod_debug_conv_occupancy(t))
p = torch.cuda.get_device_properties(0)
print(p)
hip = ctypes.CDLL('libamdhip64.so')
v = ctypes.c_int()
for name, attr in (('MaxSharedMemoryPerMultiprocessor', 74), ('MaxSharedMemoryPerBlock', 8)):
    pass
print('shared_memory_per_block', getattr(p, 'shared_memory_per_block', None), 'per_mp', getattr(p, 'shared_memory_per_multiprocessor', None))
