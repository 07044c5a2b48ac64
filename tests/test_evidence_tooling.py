"""The evidence scripts (scripts/summarize_pmc.py) label what they summarise from the kernel NAMES rocprofv3 reports: a template
that grows an argument must not silently relabel the dominant kernel (it did, twice).  CPU only."""
import csv
import io
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import summarize_pmc  # noqa: E402


def _names(csv_path):
    with open(csv_path) as f:
        return [r["Name"] for r in csv.DictReader(f)]


def test_arithmetic_labels_of_the_committed_kernel_names():
    """every matrix-kernel name of the round-4 batch-1 trace gets the arithmetic its template arguments say, by position"""
    names = _names(os.path.join(ROOT, "profiles", "r04_b1_serial_rocprofv3_kernel_stats.csv"))
    dma = [n for n in names if "conv_dma_kernel" in n]
    igemm = [n for n in names if "conv_igemm_kernel" in n]
    assert dma and igemm
    for n in dma:
        args = summarize_pmc.template_args(n)
        assert len(args) == 8                                     # <BM, BK, WAVES_K, S, BALANCED, WAVES_N, NPL, CHAN>
        want = {"2": "fp16x2 (LDS-DMA)", "3": "bf16x3 (LDS-DMA)"}[args[6]]
        assert summarize_pmc.arith(n) == want, n
    for n in igemm:
        args = summarize_pmc.template_args(n)
        want = {"0": "f32", "1": "bf16x3", "2": "fp16x2"}[args[7]]
        assert summarize_pmc.arith(n) == want, n
    # the dominant kernel of that profile is the fp16x2 LDS-DMA tile - the label the round-4 summaries got wrong
    assert summarize_pmc.arith(names[0]) == "fp16x2 (LDS-DMA)"
    assert any(summarize_pmc.arith(n) == "bf16x3 (LDS-DMA)" for n in dma)


def test_arithmetic_label_survives_a_longer_and_a_shorter_template_list():
    assert summarize_pmc.arith("conv_dma_kernel<128, 32, 2, 3, false, 1, 2, false, 7, true>") == "fp16x2 (LDS-DMA)"
    assert summarize_pmc.arith("conv_dma_kernel<128, 32, 2, 3, false, 1, 2>") == "fp16x2 (LDS-DMA)"
    assert summarize_pmc.arith("conv_dma_kernel<128, 16, 1, 4, false>") == "bf16x3 (LDS-DMA)"          # defaults: NPL = 3
    assert summarize_pmc.arith("conv_igemm_kernel<64, 64, 32, 32, 5, 1, 32, 2, 9>") == "fp16x2"
    assert summarize_pmc.arith("conv_igemm_kernel<64, 64, 32, 32, 4>") == "f32"                        # defaults: PREC = 0
    assert summarize_pmc.arith("void stem_kernel(StemParams)") == "fp16x2 (one-launch stem)"
    assert summarize_pmc.arith("void bottleneck_kernel<10>(BParams)") == "fp16x2 (one-launch bottleneck)"
    with pytest.raises(ValueError):
        summarize_pmc.arith("conv_dma_kernel<128, 32, 2, 3, false, 1, 5, false>")                      # an arithmetic nobody knows


def _write_pass(d, kernels, counters):
    """a rocprofv3 --pmc output directory with two forwards of `kernels` (each opened by the stem kernel)"""
    os.makedirs(os.path.join(d, "box"), exist_ok=True)
    kt = open(os.path.join(d, "box", "1_kernel_trace.csv"), "w", newline="")
    cc = open(os.path.join(d, "box", "1_counter_collection.csv"), "w", newline="")
    wk, wc = csv.writer(kt), csv.writer(cc)
    wk.writerow(["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
    wc.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
    i, t = 1, 1000
    for _fwd in range(4):
        for k in kernels:
            wk.writerow([i, k, t, t + 30000])
            for c, v in counters.items():
                wc.writerow([i, k, c, v])
            i += 1
            t += 40000
    kt.close()
    cc.close()


def test_summary_runs_end_to_end_and_labels_the_lds_dma_kernel(tmp_path):
    ks = ["void (anonymous namespace)::stem_kernel((anonymous namespace)::StemParams)",
          "void (anonymous namespace)::conv_dma_kernel<128, 32, 2, 3, false, 1, 2, false>((anonymous namespace)::ConvParams)",
          "void (anonymous namespace)::conv_igemm_kernel<64, 64, 32, 32, 5, 1, 32, 2>((anonymous namespace)::ConvParams)",
          "void (anonymous namespace)::conv_dma_kernel<256, 16, 1, 4, false, 1, 3, false>((anonymous namespace)::ConvParams)",
          "void (anonymous namespace)::roi_pool_avg_kernel((anonymous namespace)::RoiParams)"]
    sq, fe, wr = str(tmp_path / "sq"), str(tmp_path / "fetch"), str(tmp_path / "write")
    _write_pass(sq, ks, {"SQ_VALU_MFMA_BUSY_CYCLES": 1.0e7, "SQ_WAVE_CYCLES": 5.0e7, "SQ_WAIT_INST_ANY": 1.0e7,
                         "SQ_LDS_BANK_CONFLICT": 10.0, "SQ_LDS_IDX_ACTIVE": 1000.0})
    _write_pass(fe, ks, {"FETCH_SIZE": 10000.0, "GRBM_GUI_ACTIVE": 5.0e5})
    _write_pass(wr, ks, {"WRITE_SIZE": 8000.0})
    out = io.StringIO()
    with redirect_stdout(out):
        summarize_pmc.main(["Round t (b1)", sq, fe, wr])
    text = out.getvalue()
    rows = [ln for ln in text.splitlines() if ln.startswith("| `conv_dma_kernel<128, 32, 2, 3, false, 1, 2, false>`")]
    assert rows and "| fp16x2 (LDS-DMA) |" in rows[0]
    rows3 = [ln for ln in text.splitlines() if ln.startswith("| `conv_dma_kernel<256, 16, 1, 4, false, 1, 3, false>`")]
    assert rows3 and "| bf16x3 (LDS-DMA) |" in rows3[0]
    assert "All fp16x2 (LDS-DMA) conv dispatches together" in text
    assert "All bf16x3 (LDS-DMA) conv dispatches together" in text
    assert "roi_pool_avg_kernel" in text
