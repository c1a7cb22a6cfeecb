"""CPU: the measurement plumbing of bench.py / tools/pmc_summary.py that runs in the driver's benchmark command -- the PMC summary arithmetic
(MI355X_MICROARCH.md corrections), the fallback when rocprofv3 is not there, and the child-leg runner's failure paths."""
import csv
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _write_pass(d, name, rows):
    os.makedirs(os.path.join(d, name), exist_ok=True)
    with open(os.path.join(d, name, "bench_counter_collection.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for r in rows:
            w.writerow(dict(zip(w.fieldnames, r)))


def test_pmc_summary_arithmetic(tmp_path):
    """FETCH_SIZE is in KiB and counts 64 B per 128-byte request on gfx950 (x2), WRITE_SIZE in KiB as is; per-step = / forwards; kernels are
    grouped into families by their demangled names; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
    import pmc_summary
    d = str(tmp_path)
    conv = "void (anonymous namespace)::conv_igemm6_kernel<1, 1>(lt::ConvArgs)"
    unp = "(anonymous namespace)::unproject_qn_kernel<1, true>((anonymous namespace)::UnprojArgs) [clone .kd]"
    sa = "void (anonymous namespace)::sa3_partial_planar_kernel((anonymous namespace)::SA3Args)"
    _write_pass(d, "pmc_fetch", [(conv, "FETCH_SIZE", 1000.0)] * 3 + [(unp, "FETCH_SIZE", 10.0)] * 3 + [(sa, "FETCH_SIZE", 300.0)] * 3)
    _write_pass(d, "pmc_write", [(conv, "WRITE_SIZE", 500.0)] * 3 + [(unp, "WRITE_SIZE", 200.0)] * 3 + [(sa, "WRITE_SIZE", 0.0)] * 3)
    _write_pass(d, "pmc_mfma", [(conv, "SQ_VALU_MFMA_BUSY_CYCLES", 1024 * 1000.0 * 0.25), (conv, "GRBM_GUI_ACTIVE", 8 * 1000.0), (conv, "SQ_WAVE_CYCLES", 100.0),
                                (conv, "SQ_WAIT_ANY", 40.0), (conv, "SQ_WAIT_INST_ANY", 30.0), (conv, "SQ_ACTIVE_INST_ANY", 20.0)])
    r = pmc_summary.summarise(d, 3, batch=32)
    assert r["conv_family_bytes_per_step"] == pytest.approx(1000 * 1024 * 2 + 500 * 1024)
    assert r["hbm_kernels_bytes_per_step"]["unproject"] == pytest.approx(10 * 1024 * 2 + 200 * 1024)
    assert r["hbm_kernels_bytes_per_step"]["softargmax3d"] == pytest.approx(300 * 1024 * 2)
    assert r["conv_family_mfma_busy_frac"] == pytest.approx(0.25)
    k = r["per_kernel"]["conv_igemm6_kernel<1, 1>"]
    assert k["family"] == "conv" and k["launches_per_step"] == 1.0 and k["wave_wait_any_frac"] == pytest.approx(0.4)
    assert "unproject_qn_kernel<1, true>" in r["per_kernel"]
    # a missing pass leaves the traffic empty instead of inventing numbers
    r2 = pmc_summary.summarise(os.path.join(d, "nothing_here"), 3)
    assert r2["conv_family_bytes_per_step"] is None and r2["hbm_kernels_bytes_per_step"] == {}


def test_bench_pmc_leg_and_sub_leg_fail_soft(monkeypatch, tmp_path):
    import bench
    monkeypatch.setattr("shutil.which", lambda name: None)
    monkeypatch.setattr(os.path, "exists", lambda p, _e=os.path.exists: False if p == "/opt/rocm/bin/rocprofv3" else _e(p))

    class A:
        batch, views, volume, image, layers, dtype = 32, 4, 64, 384, 152, "bf16"
    assert bench.pmc_leg(A()) is None                          # no rocprofv3: the caller falls back to the committed file and says so
    bad = bench.sub_leg(["--no-such-flag"], 60)                # a child that exits non-zero: an error record, not an exception
    assert "error" in bad and bad["argv"] == ["--no-such-flag"]
    ok = bench.sub_leg(["--gpus", "1", "--steps", "2", "--warmup", "0", "--batch", "3", "--stub-cpu"], 120)
    assert ok.get("metric") == "stub" and ok["total_samples"] == 6 and ok["leg_wall_s"] > 0


def test_compact_line_fits_the_drivers_tail_and_keeps_every_headline_figure():
    """VERDICT r4 "next" 3: the driver stores an 8 KB tail of stdout and round 4's line was 14 KB.  bench.compact_line() of a FULL record (round 4's own line,
    profiles/r04_bench_bf16.json, plus the legs added this round) must stay within 4 KB and still carry: headline, roofline, roofline_hbm, cpu_baseline,
    parity (max abs mm, MPJPE, meets_gate), fp32_parity_mode (value, frac, meets_gate), batch_sweep["32"], one-number summaries of every leg."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_bf16.json")))
    full["train_trajectory"] = {"steps": 20, "config": {"per_gpu_batch": 4}, "curves": {k: [53.0 - 0.5 * i for i in range(20)] for k in ("fp32", "act16", "fp8v2v")},
                                "final": {"fp32": 43.5, "act16": 43.6, "fp8v2v": 43.9}, "final_rel_to_fp32": {"fp32": 0.0, "act16": 0.0023, "fp8v2v": 0.0092},
                                "max_rel_gap_to_fp32_over_the_run": {"fp32": 0.0, "act16": 0.004, "fp8v2v": 0.011}}
    full["train_mixed"]["kernel_share"] = {"conv fwd+dgrad": 0.31, "batchnorm+sums": 0.27, "wgrad": 0.25, "params+casts": 0.05, "unproject": 0.04, "other": 0.08}
    full["roofline_hbm"]["unproject"]["valu_issue_frac"] = 0.71
    full["roofline_hbm"]["unproject"]["lane_insts_per_voxel"] = 1650.0
    c = bench.compact_line(full)
    line = json.dumps(c)
    assert len(line) <= 4096, len(line)          # "keep the final line <= 4 KB"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert c[k] == full[k] or (isinstance(full[k], float) and abs(c[k] - full[k]) <= 1e-5 * abs(full[k])), k
    assert abs(c["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-3 and c["roofline"]["traffic"] and c["roofline"]["bound"] == "mfma"
    assert c["roofline_hbm"]["unproject"]["valu_issue_frac"] == 0.71 and c["roofline_hbm"]["softargmax3d"]["frac"] > 0
    assert c["parity"]["meets_gate"] is False and c["parity"]["max_abs_mm"] > 1 and c["parity"]["mpjpe_mm"] > 1
    assert c["fp32_parity_mode"]["parity"]["meets_gate"] is True and c["fp32_parity_mode"]["value"] > 100 and c["fp32_parity_mode"]["frac"] > 0.5
    assert c["batch_sweep"]["32"] > 1000 and c["config4"]["parity_fp32"]["meets_gate"] is True and c["config4"]["unproject"]["frac"] > 0
    assert c["train_mixed"]["traffic_gb"] > 100 and c["train_mixed"]["kernel_share"]["batchnorm+sums"] == 0.27 and c["train_fp8v2v"]["value"] > 0
    assert c["train_trajectory"]["final"]["fp8v2v"] == 43.9 and c["cpu_baseline"]["cores"] == 32
    # a failed leg stays visible as an error, not as a crash of the line
    full["config4"] = {"error": "timed out after 600 s", "argv": []}
    assert bench.compact_line(full)["config4"] == {"error": "timed out after 600 s"}


def test_clock_sampler_counts_only_reads_inside_the_timed_window(tmp_path):
    """Round 6: bench.ClockSampler -- rocm-smi is polled by a shell loop that stamps every read before (T) and after (E); window(t0, t1) averages the reads
    that BEGAN and ENDED inside the timed region (a read that ends after the last step sees the idling chip), survives a missing rocm-smi / an empty log,
    and the two figures ride on the compact line next to roofline.frac."""
    import bench
    s = bench.ClockSampler.__new__(bench.ClockSampler)
    s.proc, s.path = None, str(tmp_path / "clk.log")

    def read(t, e, mhz, w):
        return "T %.3f\nGPU[0]\t\t: sclk clock level: 1: (%dMhz)\nGPU[0]\t\t: Current Socket Graphics Package Power (W): %.1f\nE %.3f\n" % (t, mhz, w, e)
    open(s.path, "w").write(read(99.0, 99.2, 2400, 300.0) +           # before the window (clocks not settled)
                            read(100.1, 100.3, 2100, 1370.0) + read(100.5, 100.7, 2080, 1366.0) +
                            read(100.9, 101.15, 900, 400.0) +        # began inside, ended after the last step: idle chip
                            "T 101.3\n")                             # the loop was killed mid-read
    r = s.window(100.0, 101.0)
    assert r == {"sclk_mhz": 2090, "power_w": 1368, "samples": 2}
    assert not os.path.exists(s.path)                                 # the temporary log is removed
    s2 = bench.ClockSampler.__new__(bench.ClockSampler)
    s2.proc, s2.path = None, str(tmp_path / "empty.log")
    open(s2.path, "w").write("T 100.2\nE 100.4\n")                    # rocm-smi printed nothing (no such device)
    assert s2.window(100.0, 101.0) is None
    s3 = bench.ClockSampler.__new__(bench.ClockSampler)
    s3.proc, s3.path = None, None                                     # rocm-smi not installed
    assert s3.window(0.0, 1.0) is None
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_bf16_full_record.json")))
    full["roofline"].update(sclk_mhz=2090, power_w=1368)
    c = bench.compact_line(full)
    assert c["roofline"]["sclk_mhz"] == 2090 and c["roofline"]["power_w"] == 1368 and len(json.dumps(c)) <= 4096
