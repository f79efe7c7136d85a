"""bench.py's final stdout line must stay far below the ~8 KB of tail the driver keeps (round 3's 29 KB line left
BENCH_r03.parsed null): `bench.compact_line` is bounded on a worst-case result, keeps the contract's keys, `roofline` and
`cpu_baseline`, and `bench.emit` prints nothing to stdout after it (the full result goes to bench_detail.json + stderr)."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")
SYM = "gemm_f32x3_ws_kernel<false, false, false>" + "x" * 40  # longer than any real template instantiation


def _roofline(fat=True):
    rf = {"bound": "mfma", "achieved": 171.23, "peak": 416.7, "unit": "TFLOP/s", "frac": 0.4109, "traffic": 193456789.0,
          "traffic_source": "profiles/pmc_gemm.json: " + "s" * 300, "kernel": SYM, "avg_launch_us": 94.3, "launches_per_step": 49,
          "measured": "m" * 300, "flops_per_launch_avg": 1.234e10,
          "all_gemm_kernels": {"ms_per_step": 12.345, "tflops": 151.2, "frac": 0.3628, "executed_tflop_per_step": 1.9191}}
    if fat:
        rf["per_kernel"] = [{"kernel": SYM, "launches_per_step": 48, "avg_us": 100.1, "tflops": 150.3}] * 8
        rf["per_shape"] = [{"kernel": "gemm_f32x3_ws_kernel", "M": 4096, "N": 3072, "K": 768, "splits": 1, "launches_per_step": 12,
                            "avg_us": 110.0, "tflops": 175.5}] * 12
    return rf


def _worst_case():
    sec = {}
    for k in ("c1_fp32", "c1_fp32_graph", "c2_fp32_pipe", "c3_bf16", "c4_bf16", "c5_extra", "c6_extra"):
        sec[k] = {"config": k, "value": 12345.67, "unit": "sentences/s", "ms_per_step": 123.456, "steps": 40, "dtype": "d" * 250,
                  "dtype_short": "fp32x3", "workload": "w" * 300, "mfma_fraction_of_step": 0.1234, "loss": 102.2119,
                  "tolerance": "t" * 250, "accuracy": "a" * 300, "roofline": _roofline()}
    sec["broken"] = {"error": "RuntimeError(" + "e" * 500 + ")"}
    return {"metric": "training sentences/sec (fwd+bwd)", "value": 2345.67, "unit": "sentences/s", "n_gpus": 8, "steps": 20, "warmup": 5,
            "ms_per_step": 13.642, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32 (3xbf16 split products, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "W" * 420, "global_batch": 256, "seq_len": 128, "prefix": 36,
                       "parallelism": "dp8 (RCCL all-reduce overlapped with backward)"},
            "median_ms_per_step": 13.6, "value_median": 2350.0, "loss": 102.2119, "mfma_fraction_of_step": 0.3803,
            "mfma_fraction_of_step_executed": 0.35, "peak_tflops": 416.7, "flop_per_sentence_train": 67557285888,
            "fwd_bwd_without_optimizer": {"value": 2400.0, "ms_per_step": 13.3}, "real_token_rows": 0.5771,
            "flop_per_sentence_train_real_rows": 38600000000, "note_flops": "n" * 500, "padding": "p" * 80,
            "padded": {"value": 2200.0, "ms_per_step": 14.5, "mfma_fraction_of_step_algorithmic": 0.36, "note": "n" * 300},
            "full_length": {"value": 1800.0, "ms_per_step": 17.76},
            "n_ranks_seen": 8, "backend": "nccl", "rank_ms_spread": 0.012,
            "grad_sync": {"wire": "bf16", "buckets": 4, "comm_stream_ms_per_step": 1.234, "exposed_tail_ms_per_step": 0.123, "note": "g" * 300},
            "roofline": _roofline(), "roofline_fp32_pipe": _roofline(), "value_fp32_pipe": 1723.0, "ms_per_step_fp32_pipe": 18.57,
            "secondary": sec,
            "cpu_baseline": {"value": 6.498, "unit": "sentences/s", "cores": 64, "kind": "port", "sample": "s" * 400, "seconds_per_step": 4.9},
            "cpu_baseline_c1": {"value": 5.8, "unit": "sentences/s", "cores": 64, "kind": "port", "sample": "s" * 400}}


def test_compact_line_is_bounded_and_keeps_the_contract():
    import bench
    res = _worst_case()
    assert len(json.dumps(res)) > 20000  # the full result really is the size that broke round 3
    line = bench.compact_line(res)
    text = json.dumps(line)
    assert len(text) < 4096 == bench.LINE_LIMIT
    for k in CONTRACT:
        assert k in line, k
        if k != "config":
            assert line[k] == res[k]
    for k in ("workload", "global_batch", "seq_len", "parallelism"):
        assert k in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert rf[k] == res["roofline"][k]
    assert "per_kernel" not in rf and "per_shape" not in rf
    cb = line["cpu_baseline"]
    assert {k: cb[k] for k in ("value", "unit", "cores", "kind")} == {"value": 6.498, "unit": "sentences/s", "cores": 64, "kind": "port"}
    assert "sample" in cb
    assert line["n_ranks_seen"] == 8 and line["backend"] == "nccl"
    assert res["config"]["workload"] == "W" * 420  # the full result is not modified by the cut


def test_compact_line_of_a_typical_result_keeps_the_secondaries():
    import bench
    res = _worst_case()
    res["config"]["workload"] = "TVNetSAModel2 BERT-base random-init, fwd+bwd+AdamW(HIP, overlapped with backward), bs=32/GPU"
    res["cpu_baseline"]["sample"] = "median of 9 fwd+bwd steps of the bench workload itself on torch CPU, 64 threads"
    for k in ("c5_extra", "c6_extra", "broken"):
        res["secondary"].pop(k)
    for v in res["secondary"].values():
        v["roofline"]["kernel"] = "gemm_bf16x_kernel<128, 96, 4, 1, false, true, 2, false>"
    line = bench.compact_line(res)
    assert len(json.dumps(line)) < 4096
    assert set(line["secondary"]) == set(res["secondary"])
    for v in line["secondary"].values():
        assert set(v) == {"value", "ms_per_step", "dtype", "roofline_frac", "roofline_kernel"}
    assert line["grad_sync"]["wire"] == "bf16" and "note" not in line["grad_sync"]
    # the padded and full-length runs stay beside `value` (round-4 review, gate 3b)
    assert line["padded"] == {"value": 2200.0, "ms_per_step": 14.5} and line["full_length"] == {"value": 1800.0, "ms_per_step": 17.76}


def test_emit_prints_the_compact_line_last_and_the_detail_elsewhere(tmp_path):
    import bench
    res = _worst_case()
    out, err = io.StringIO(), io.StringIO()
    detail = tmp_path / "bench_detail.json"
    bench.emit(res, out=out, err=err, detail_path=str(detail))
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096  # ONE line on stdout, nothing after it
    parsed = json.loads(lines[0])
    assert parsed["value"] == res["value"] and parsed["roofline"]["frac"] == res["roofline"]["frac"]
    assert json.loads(detail.read_text()) == res  # everything else survives in the detail file ...
    assert "per_shape" in err.getvalue()  # ... and on stderr
