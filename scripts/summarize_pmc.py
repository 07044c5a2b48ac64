#!/usr/bin/env python3
"""Markdown summary of the three rocprofv3 --pmc passes of bench.py (SQ counters, FETCH_SIZE+GRBM, WRITE_SIZE).
Usage: summarize_pmc.py <pmc_sq dir> <pmc_fetch dir> <pmc_write dir>"""
import csv, glob, collections, re, sys

def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:60]

def load(d):
    cc = sorted(glob.glob(d + '/*/*counter_collection.csv'))[-1]
    kt = sorted(glob.glob(d + '/*/*kernel_trace.csv'))[-1]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(cc)):
        per[short(r['Kernel_Name'])][r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    return per, dur

sq, dsq = load(sys.argv[1]); fe, dfe = load(sys.argv[2]); wr, dwr = load(sys.argv[3])
print("# Round 1 - PMC counters, `bench.py --no-graph` (B=1, 3x800x1333, ResNet-50), rocprofv3 --pmc, separate passes\n")
print("Per dispatch, averaged over the second half of each kernel's dispatches (the timed steps; the first half contains")
print("plan building). `FETCH_SIZE` / `WRITE_SIZE` are in KiB; on gfx950 `FETCH_SIZE` under-reports wide coalesced reads by 2x")
print("(MI355X_MICROARCH.md, HBM section), so HBM read bytes = 2 x FETCH_SIZE.  Commands: profiles/README.md.\n")
g = [c.get('GRBM_GUI_ACTIVE') for k in fe if 'conv_igemm' in k for _, c in fe[k].items() if c.get('GRBM_GUI_ACTIVE')]
d = [dfe[i] for k in fe if 'conv_igemm' in k for i, c in fe[k].items() if c.get('GRBM_GUI_ACTIVE')]
clock_meas = sum(g) / 8 / (sum(d) * 1e-6)
clock = min(clock_meas, 2.4e9)   # GRBM_GUI_ACTIVE/8/duration reads high on dispatches < 0.3 ms (microarch guide); cap at f_max
print("## Matrix-core utilisation of the conv GEMMs (pass 1: SQ counters)\n")
print("`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x 2.4 GHz) (clock check from pass 2 below).\n")
print("| kernel | dispatches | avg us | MFMA busy | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_LDS_BANK_CONFLICT |")
print("|---|---:|---:|---:|---:|---:|---:|")
tot_mf = tot_cyc = 0
for k in sorted(sq):
    if 'conv_igemm' not in k:
        continue
    ds = list(sq[k].items()); ds = ds[len(ds) // 2:]
    n = len(ds)
    us = sum(dsq[i] for i, _ in ds) / n
    mf = sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for _, c in ds) / n
    wc = sum(c.get('SQ_WAVE_CYCLES', 0) for _, c in ds) / n
    wi = sum(c.get('SQ_WAIT_INST_ANY', 0) for _, c in ds) / n
    wa = sum(c.get('SQ_WAIT_ANY', 0) for _, c in ds) / n
    bc = sum(c.get('SQ_LDS_BANK_CONFLICT', 0) for _, c in ds) / n
    cyc = 4 * 256 * us * 1e-6 * clock
    tot_mf += mf * n; tot_cyc += cyc * n
    print(f"| `{k}` | {n} | {us:.1f} | {100 * mf / cyc:.1f}% | {wi / wc:.2f} | {wa / wc:.2f} | {bc:.0f} |")
print(f"\nAll conv_igemm dispatches together: MFMA pipes busy **{100 * tot_mf / tot_cyc:.1f}%** of the time these kernels run (this counts the")
print("zero-weight padding work of the 7x8x4 stem and of partial edge tiles, which the algorithmic TFLOP/s in bench.py does not).\n")
print("## HBM traffic of the memory-bound kernels (passes 2 and 3: FETCH_SIZE, WRITE_SIZE)\n")
print("| kernel | avg us | FETCH_SIZE KB | WRITE_SIZE KB | HBM MB = (2*FETCH+WRITE)*1024/1e6 | achieved GB/s (PMC bytes / time) | algorithmic MB (DESIGN.md section 4) |")
print("|---|---:|---:|---:|---:|---:|---:|")
alg = {'maxpool3x3s2_kernel': 85.4, 'nchw_to_nhwc_small_kernel': 29.9, 'rpn_decode_kernel': 0.42, 'roi_pool_avg_kernel': 11.1,
       'nms_mask_kernel<4, -1>': 1.18, 'nms_scan_kernel<1>': 1.13, 'sort_topk_kernel<10>': 0.11, 'detections_kernel': 0.5}
for k in ['nchw_to_nhwc_small_kernel', 'maxpool3x3s2_kernel', 'conv_reduce_kernel', 'rpn_decode_kernel', 'sort_topk_kernel<10>',
          'nms_mask_kernel<4, -1>', 'nms_scan_kernel<1>', 'roi_pool_avg_kernel', 'detections_kernel']:
    if k not in fe:
        continue
    ds = list(fe[k].items()); ds = ds[len(ds) // 2:]
    n = len(ds); us = sum(dfe[i] for i, _ in ds) / n
    f = sum(c.get('FETCH_SIZE', 0) for _, c in ds) / n
    dw = list(wr[k].items()); dw = dw[len(dw) // 2:]
    w = sum(c.get('WRITE_SIZE', 0) for _, c in dw) / max(1, len(dw))
    mb = (2 * f + w) * 1024 / 1e6
    a = alg.get(k)
    print(f"| `{k}` | {us:.1f} | {f:.0f} | {w:.0f} | {mb:.2f} | {mb * 1e6 / (us * 1e-6) / 1e9:.0f} | {'' if a is None else a} |")
print(f"\nClock check: sum(GRBM_GUI_ACTIVE)/8/sum(duration) over the conv dispatches = {clock_meas / 1e9:.2f} GHz (this quotient reads high on")
print("dispatches shorter than ~0.3 ms; the table uses min(measured, 2.4 GHz)): the chip holds its full clock under f32 MFMA load.")


def last_forward(per):
    """Counter sums per kernel family over the dispatches of the last forward (from the last NCHW->NHWC launch on)."""
    start = max(int(i) for k in per if 'nchw_to_nhwc' in k for i in per[k])
    out = collections.defaultdict(lambda: [0, 0.0])
    for k in per:
        fam = 'conv_igemm_kernel' if 'conv_igemm' in k else k
        for i, c in per[k].items():
            if int(i) >= start:
                out[fam][0] += 1
                out[fam][1] += sum(v for n, v in c.items() if n in ('FETCH_SIZE', 'WRITE_SIZE'))
    return out

lf, lw = last_forward(fe), last_forward(wr)
print("\n## HBM traffic of the conv GEMMs in one forward (last forward of passes 2 and 3)\n")
print("| kernel family | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM MB = (2*FETCH+WRITE)*1024/1e6 |")
print("|---|---:|---:|---:|---:|")
tot = 0.0
for fam in ('conv_igemm_kernel', 'conv_reduce_kernel'):
    if fam not in lf:
        continue
    mb = (2 * lf[fam][1] + lw[fam][1]) * 1024 / 1e6
    tot += mb
    print(f"| `{fam}` | {lf[fam][0]} | {lf[fam][1]:.0f} | {lw[fam][1]:.0f} | {mb:.1f} |")
print(f"\nTogether {tot / 1e3:.2f} GB per forward (B=1, all 57 GEMM launches incl. RPN and head).  bench.py's `roofline.traffic` is the")
print("same measurement made live (two `rocprofv3 --pmc` child passes) over the 53 trunk launches with that run's own tile")
print("choices, per launch; `roofline.algorithmic_bytes_per_launch` is each layer's input + output + weights (+ residual) once.")
print("The excess over the algorithmic bytes is (a) the K-slice partial slabs (written by the GEMM, read by the reduce kernel),")
print("(b) the activation tile re-read by every output-channel tile of its row block once it has left the XCD's L2,")
print("(c) the 7x8x4 stem reading its 4-channel input 7 times.  The trunk is matrix-pipe / latency bound at batch 1, not HBM bound.")
