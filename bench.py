#!/usr/bin/env python
"""Headline benchmark: training sentences/sec (fwd+bwd) of the MTVAF hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # no launcher: this process (which has made no GPU call) starts the N ranks itself

Workload (BASELINE.json configs[1]): TVNetSAModel2 (BERT-base, random init N(0,0.02)), fp32, per-GPU
batch 32, seq_len 128, 36 visual prefix slots (main image + 8 aux crops through the prompt generator),
Twitter-shaped synthetic batch, train mode (all 37+ dropout sites live), one step = forward (incl. CRF
Viterbi decode, as the reference forward does) + loss.backward() + AdamW step.  Weak scaling: per-GPU
work is fixed; N > 1 adds the RCCL gradient all-reduce (overlapped with backward) inside the timed region.

Since round 5 the timed run is the library default, padding-free execution (the encoder layers run on the packed unmasked
token rows; loss, decoded tags and parameter gradients are those of the padded run: tests/test_unpad_gpu.py, the `pad_mode`
fixture): `config.workload` says so, every roofline / fraction in the line counts EXECUTED flops only (never SURVEY 8d's
F_train, which credits masked rows nobody computes), and the line carries the `padded` run (--padded / MTVAF_UNPAD=0) and the
`full_length` run (nothing to skip) beside `value`.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0,  # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
               "fp32x3": 2500.0 / 6}  # fp32 products as six bf16 MFMA products (csrc/gemm_f32x3.hip): fp32-equivalent peak
LABELS = ["O", "B-NEU", "I-NEU", "B-POS", "I-POS", "B-NEG", "I-NEG", "X", "[CLS]", "[SEP]"]


def f_fwd(S, P, H=768, L=12, C=11):
    """Algorithmic forward FLOPs per sentence."""
    return L * (24 * S * H * H + 4 * S * (S + P) * H) + 2 * S * H * C


def synthetic_batch(B, S, n_aux, vocab, seed, device, full_length=False):
    """Twitter-shaped synthetic batch."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.full((B,), S) if full_length else torch.randint(16, S + 1, (B,), generator=g)
    lens[0] = S
    ids = torch.randint(1000, vocab, (B, S), generator=g)
    labels = torch.randint(1, 11, (B, S), generator=g)
    mask = (torch.arange(S)[None, :] < lens[:, None]).long()
    ids[:, 0] = 101
    ids[torch.arange(B), lens - 1] = 102
    ids = ids * mask
    labels = labels * mask
    labels[:, 0] = 9
    feats = torch.randn(B, 3840, 2, 2, generator=g).abs()
    aux = torch.randn(B, n_aux, 3840, 2, 2, generator=g).abs()
    tt = torch.zeros_like(ids)
    return tuple(t.to(device) for t in (ids, mask, tt, labels, feats, aux))


def build_model(device, arch="bert", max_pos=512):
    from transformers import BertConfig, RobertaConfig
    from mtvaf_amd.models.bert_model import TVNetSAModel2
    if arch == "roberta":  # roberta-base architecture (BASELINE config 3)
        cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=max(514, max_pos + 2), type_vocab_size=1,
                            layer_norm_eps=1e-5, pad_token_id=1)
    else:
        cfg = BertConfig(max_position_embeddings=max(512, max_pos))  # bert-base-uncased, dropout 0.1
    args = types.SimpleNamespace(bert_name="roberta-base" if arch == "roberta" else "bert-base-uncased",
                                 bert_config=cfg, use_prefix=True, vao=False,
                                 noauxloss=True, use_probe=False, n_gpu=1, alpha=0.0, prefix_len=4, prefix_dim=768,
                                 device=device, resnet_root=None, use_152=False)
    torch.manual_seed(1234)
    return TVNetSAModel2(LABELS, None, args).to(device), cfg


def cpu_baseline(B, S, n_aux, seconds_budget=40.0, min_steps=3):
    """The reference algorithm on the host cores: the CPU oracle (a line-by-line restatement of the reference modules,
    proven equal to them by tests/test_oracle_golden.py) doing fwd+bwd of the SAME workload as the GPU line: the full
    batch, prompt generator (`O.visual_prompt`, 1 + n_aux region-feature images) -> encoder -> fc -> CRF decode + NLL.
    Median of >= `min_steps` steps within the time budget."""
    from oracle import mtvaf_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import params as PR
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads: B={B} S={S} aux={n_aux}")
    cfg = PR.BASE_BERT
    sd = {**{"bert." + k: v.requires_grad_(True) for k, v in PR.encoder_params(cfg, 1, std=0.02).items()},
          **{k: v.requires_grad_(True) for k, v in PR.head_params(cfg, 2).items()},
          **{k: v.requires_grad_(True) for k, v in PR.prompt_params(5).items()}}
    ids, mask, tt, labels = PR.text_batch(cfg, 3, B, S, lo_id=1000)
    labels[:, 0] = 9
    g = torch.Generator().manual_seed(7)
    feats = torch.randn(B, 4, 3840, generator=g).abs()
    aux = [torch.randn(B, 4, 3840, generator=g).abs() for _ in range(n_aux)]

    def step():
        pkv, _, _ = O.visual_prompt(sd, feats, aux, num_layers=cfg.layers, num_heads=cfg.heads)
        loss, _, _, _ = O.tvnet2_forward(sd, ids, mask, tt, labels, pkv, cfg.layers, cfg.heads, cfg.eps)
        loss.backward()
        for v in sd.values():
            v.grad = None

    t_all = time.perf_counter()
    step()  # warm-up (thread pool, allocator)
    times = []
    while len(times) < min_steps or (time.perf_counter() - t_all < seconds_budget and len(times) < 9):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": round(B / med, 3), "unit": "sentences/s", "cores": cores, "kind": "port",
            "sample": f"median of {len(times)} fwd+bwd steps of the bench workload itself (B={B}, S={S}, P={4 * (1 + n_aux)}: "
                      f"prompt generator + BERT-base encoder + fc + CRF decode/NLL, fp32) on torch CPU, {cores} threads",
            "seconds_per_step": round(med, 3)}


def frontend_cache_throughput(device, batch=32, n_aux=3, hw=224, reps=3, compute="fp32"):
    """Row f1 (the step before the path): images/s of building the region-feature cache -- the frozen ResNet-50 pyramid of
    `ImageModel` (reference: models/bert_model.py:63-111; torch / MIOpen, random weights: the reference's .pth files are not
    in the image) run once per image in inference mode by `RegionFeatureCache.extract` on `batch` sentences of 1 + n_aux
    images of hw x hw.  A detail figure (bench_detail.json), never part of `value`."""
    from mtvaf_amd.features import RegionFeatureCache
    from mtvaf_amd.models.bert_model import ImageModel
    torch.manual_seed(5)
    im = ImageModel(resnet_root="random").to(device).eval()
    cache = RegionFeatureCache(im, compute=compute)
    x = torch.randn(batch, 3, hw, hw, device=device)
    aux = torch.randn(batch, n_aux, 3, hw, hw, device=device)
    cache.extract(x, aux)  # (MIOpen picks its algorithms here)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        feats, fa = cache.extract(x, aux)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    n_img = batch * (1 + n_aux)
    return {"images_per_s": round(n_img / dt, 1), "sentences_per_s": round(batch / dt, 1), "ms_per_batch": round(1e3 * dt, 2),
            "what": f"RegionFeatureCache.extract, ResNet-50 pyramid (random weights), {batch} sentences x (1 + {n_aux}) images of "
                    f"{hw}x{hw}, {'fp32' if compute == 'fp32' else 'channels-last bf16 trunk with folded BatchNorm, fp32 region pooling'}, "
                    f"eval-mode BatchNorm; output {tuple(feats.shape)} + {tuple(fa.shape)}"}


def pmc_traffic(symbol, dtype="fp32", batch=32, seq=128, unpad=True):
    """HBM-side bytes per launch of `symbol` from the committed rocprofv3 PMC passes (profiles/pmc_gemm.json, written by
    tools/pmc_to_json.py: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B/lane streams, plus
    WRITE_SIZE).  -> (bytes or None, provenance string): the counters come from a separate profiled run of this same
    command, not from the run that prints the line."""
    # one file per measured workload (bytes per launch depend on the token count): the fp32 headline, bf16 at bs 32 / bs 64
    # (round 5: pmc_gemm.json = the padding-free headline, pmc_gemm_padded.json = the padded run of rounds 1-4; the bf16 files
    # were collected on the padded layout)
    if (dtype, batch, seq) == ("fp32", 32, 128):
        name = "pmc_gemm.json" if unpad else "pmc_gemm_padded.json"
    elif unpad:
        return None, None
    else:
        name = f"pmc_gemm_{dtype}_b{batch}.json"
    path = os.path.join(ROOT, "profiles", name)
    if seq != 128:
        return None, None
    if not os.path.exists(path):
        return None, None
    try:
        d = json.load(open(path))
        rec = d.get(symbol)
        if rec is None:
            return None, None
        return rec["traffic_bytes_per_launch"], d.get("_source", os.path.basename(path) + " (rocprofv3 --pmc passes, committed)")
    except Exception:
        return None, None



def _planes_on():
    from mtvaf_amd import engine as _engine
    return bool(_engine.F32_PLANES)


def roofline_pass(eager_step, mask, B, S, dtype, unpad=False, nprof=3, peak_key=None):
    """Roofline object of the dominant GEMM kernel of `eager_step`, measured live: HIP events that the library records on the
    launch stream directly around each main GEMM kernel (mtvaf_prof_start/stop) in `nprof` further steps, with the
    weight-gradient side stream serialised so that every kernel is timed alone."""
    from mtvaf_amd import engine as _engine
    from mtvaf_amd import hip
    side_was = _engine.DW_SIDE_STREAM
    _engine.DW_SIDE_STREAM = False
    try:
        eager_step()
        torch.cuda.synchronize()
        hip.prof_start(8192)
        for _ in range(nprof):
            eager_step()
        torch.cuda.synchronize()
        recs = hip.prof_stop(8192)
    finally:
        _engine.DW_SIDE_STREAM = side_was
    by_sym, by_shape = {}, {}
    # weight-gradient launches that walk the k-tile list execute 32 rows per LISTED tile of the token axis:
    # their flops are counted from the list, not from the padded token count
    k_listed = 32 * int(((mask.view(-1, 32).sum(1) > 0).sum()).item()) if (B * S) % 32 == 0 else B * S
    for k, ms in recs:
        sym = hip.kernel_symbol(k["cfg"], k["la"], k["lb"], k["fast"])
        d = by_sym.setdefault(sym, [0.0, 0, 0.0])
        d[0] += ms
        d[1] += 1
        kk = min(k["K"], k_listed) if (k["fast"] & 8) else k["K"]
        d[2] += 2.0 * k["M"] * k["N"] * kk
        sk = (sym, k["M"], k["N"], kk, k["splits"])
        e = by_shape.setdefault(sk, [0.0, 0])
        e[0] += ms
        e[1] += 1
    def peak_of(sym_):  # the matrix pipe a kernel symbol runs on (the split mode leaves few-tile products on the fp32 pipe)
        if sym_.startswith("gemm_f32x3") or sym_.startswith("gemm_f32p16"):  # (six bf16 products per fp32 product either way)
            return PEAK_TFLOPS["fp32x3"]
        return PEAK_TFLOPS["bf16"] if sym_.startswith("gemm_bf16") else PEAK_TFLOPS["fp32"]
    tot_ms = sum(v[0] for v in by_sym.values()) / nprof
    tot_fl = sum(v[2] for v in by_sym.values()) / nprof
    tot_ideal_ms = sum(v[2] / (peak_of(k_) * 1e12) * 1e3 for k_, v in by_sym.items()) / nprof  # every launch at ITS pipe's peak
    sym, (ms, cnt, fl) = max(by_sym.items(), key=lambda kv: kv[1][0])
    avg_us = 1e3 * ms / cnt
    ach = (fl / cnt) / (avg_us * 1e-6) / 1e12
    traffic, traffic_src = pmc_traffic(sym, dtype, B, S, unpad)
    peak = peak_of(sym)
    return {
        "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src, "kernel": sym,
        "avg_launch_us": round(avg_us, 1), "launches_per_step": cnt // nprof,
        "measured": f"HIP events around each main GEMM kernel, {nprof} steps with the weight-gradient side stream serialised "
                    "(MTVAF_DW_STREAM=0) so that every kernel is timed alone",
        "flops_per_launch_avg": fl / cnt,
        "all_gemm_kernels": {"ms_per_step": round(tot_ms, 3), "tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                             "frac": round(tot_ideal_ms / tot_ms, 4),
                             "executed_tflop_per_step": round(tot_fl / 1e12, 4)},
        "per_kernel": [{"kernel": s_, "launches_per_step": c_ // nprof, "avg_us": round(1e3 * m_ / c_, 1),
                        "tflops": round(f_ / (m_ * 1e-3) / 1e12, 1)}
                       for s_, (m_, c_, f_) in sorted(by_sym.items(), key=lambda kv: -kv[1][0])[:8]],
        "per_shape": [{"kernel": k_[0].split("<")[0], "M": k_[1], "N": k_[2], "K": k_[3], "splits": k_[4],
                       "launches_per_step": c_ // nprof, "avg_us": round(1e3 * m_ / c_, 1),
                       "tflops": round(2.0 * k_[1] * k_[2] * k_[3] / (1e3 * m_ / c_ * 1e-6) / 1e12, 1)}
                      for k_, (m_, c_) in sorted(by_shape.items(), key=lambda kv: -kv[1][0])[:12]]}


def secondary_config(name, device, dtype, arch, B, S, n_aux, steps=10, warmup=3, split=False, unpad=False, graph=False):
    """One of the other BASELINE configurations measured in the same process, AFTER the headline's timed region (the
    headline's value / config / dtype are untouched): the same step (forward incl. Viterbi + backward + AdamW overlapped
    with the backward pass), `steps` timed steps bracketed by synchronisation, its own roofline object."""
    from mtvaf_amd import hip
    from mtvaf_amd.optim import AdamW
    hip.set_compute_dtype(dtype)
    from mtvaf_amd import engine as _engine
    split_was, unpad_was = hip.f32_split(), _engine.UNPAD
    hip.f32_split(split)
    _engine.UNPAD = bool(unpad)
    try:
        model, cfg = build_model(device, arch, S)
        model.train()
        opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model, overlap=not graph)
        ids, mask, tt, labels, feats, aux = synthetic_batch(B, S, n_aux, cfg.vocab_size, 1234, device)
        kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)
        gstep = None
        if graph:  # forward + backward as ONE HIP graph (mtvaf_amd.graph), the optimizer eagerly behind each replay
            from mtvaf_amd.graph import GraphedTrainStep
            gstep = GraphedTrainStep(model, kw)

        def step():
            if gstep is not None:
                out = gstep(**{k: v for k, v in kw.items() if v is not None})
            else:
                out = model(**kw)
                out.loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            assert len(out.logits) == B
            return out
        import gc
        for i in range(warmup):
            if i == warmup - 1:  # (the garbage of the configurations measured before: collected before the LAST warm-up step, see main())
                torch.cuda.synchronize()
                gc.collect()
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        P = 4 * (1 + n_aux)
        v = B * steps / dt
        res = {"config": name, "value": round(v, 2), "unit": "sentences/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
               "dtype": "fp32 (fp32 MFMA pipe)" if dtype == "fp32" else "bf16 MFMA / fp32 accumulate+storage",
               "dtype_short": ("fp32x3" if split else "fp32-pipe") if dtype == "fp32" else "bf16",
               "workload": f"TVNetSAModel2 {'RoBERTa' if arch == 'roberta' else 'BERT'}-base random-init, fwd+bwd+AdamW(HIP, overlapped), "
                           f"bs={B}, seq_len={S}, {P} visual prefix slots, train mode, ragged 16..S sequences",
               "mfma_fraction_of_step": round(v * 3 * f_fwd(S, P) / (PEAK_TFLOPS[dtype] * 1e12), 4),
               "loss": round(float(out.loss.detach()), 4)}
        if dtype != "fp32":
            res["tolerance"] = ("mixed precision: emissions <= 1.5e-2, loss <= 2e-3, >= 97 % of the decoded tags vs the fp32 oracle "
                                "(tests/test_configs_gpu.py; north_star's 1e-3 / bit-exact tags hold in fp32 mode only)")
        if graph:
            res["workload"] += ", forward + backward replayed as one HIP graph (the host-bound configuration: eager runs read 780-1035)"
            gstep.close()
            gstep = None
        if hip.streamk_errors():
            raise RuntimeError("a stream-K launch reported a timed-out wait")
        if split and dtype == "fp32":
            res["dtype"] = ("fp32 operands / results / accumulation; every fp32 product formed on the bf16 matrix pipe as the six "
                            "significant partial products of three-way bf16-split operands (each exact in fp32)")
            res["accuracy"] = ("error of the split GEMM against the fp64 product is at or below the fp32 MFMA pipe's on the same "
                               "inputs (tests/test_ops_gpu.py::test_gemm_f32_split_accuracy); the library default, so every fp32 "
                               "parity test (1e-3 vs the oracle, bit-exact tags, reference goldens) runs in this mode")
            res["mfma_fraction_of_step"] = round(v * 3 * f_fwd(S, P) / (PEAK_TFLOPS["fp32x3"] * 1e12), 4)
        if unpad:
            res["workload"] += ", padding-free execution (masked token rows not computed: DESIGN.md 7)"
            # executed flops only: the flops of the rows that are real tokens, per sentence with its own length
            f_ex = 3 * sum(f_fwd(int(n), P) for n in mask.sum(1).tolist()) / B
            res["mfma_fraction_of_step"] = round(v * f_ex / (PEAK_TFLOPS["fp32x3" if (split and dtype == "fp32") else dtype] * 1e12), 4)
        res["roofline"] = roofline_pass(step, mask, B, S, dtype, unpad=unpad, peak_key="fp32x3" if (split and dtype == "fp32") else None)
        return res
    finally:
        hip.set_compute_dtype("fp32")
        hip.f32_split(split_was)
        _engine.UNPAD = unpad_was



LINE_LIMIT = 4096  # the driver keeps ~8 KB of stdout tail: the final line stays far below it (tests/test_bench_line.py)
_HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "median_ms_per_step", "loss", "mfma_fraction_of_step", "mfma_fraction_of_step_executed",
              "value_fp32_pipe", "ms_per_step_fp32_pipe", "n_ranks_seen", "backend", "rccl_version", "rank_ms_spread",
              "rank_token_rows", "sharding")
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "launches_per_step")


def _short_roofline(rf):
    if not isinstance(rf, dict):
        return rf
    out = {k: rf[k] for k in _ROOF_KEYS if k in rf}
    if "all_gemm_kernels" in rf:
        out["all_gemm_frac"] = rf["all_gemm_kernels"].get("frac")
        out["all_gemm_ms_per_step"] = rf["all_gemm_kernels"].get("ms_per_step")
    return out


def compact_line(res):
    """The ONE stdout line of the driver's contract, cut down from the full result (which goes to bench_detail.json and to
    stderr): the contract's keys, `roofline` without its per-kernel / per-shape lists, `cpu_baseline`, and for every secondary
    configuration {value, ms_per_step, dtype, roofline_frac, roofline_kernel}.  Never longer than LINE_LIMIT bytes: optional
    parts are dropped, least important first, until it fits."""
    line = {k: res[k] for k in _HEAD_KEYS if k in res}
    if isinstance(line.get("config"), dict):
        line["config"] = dict(line["config"])
    if "roofline" in res:
        line["roofline"] = _short_roofline(res["roofline"])
    if "roofline_fp32_pipe" in res:
        line["roofline_fp32_pipe"] = _short_roofline(res["roofline_fp32_pipe"])
    if "cpu_baseline" in res:
        line["cpu_baseline"] = {k: res["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample") if k in res["cpu_baseline"]}
    if isinstance(res.get("parity"), dict):  # measured deviations from the oracle at the timed shape (tests/parity_report.py)
        line["parity"] = {k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in res["parity"].items()
                          if k in ("emissions_max_rel", "emissions_elem_rel_p99", "emissions_elem_rel_max", "loss_rel", "tags_equal",
                                   "worst_grad_rel", "error")}
    if isinstance(res.get("grad_sync"), dict):
        line["grad_sync"] = {k: v for k, v in res["grad_sync"].items() if k != "note"}
    if isinstance(res.get("padded"), dict):
        line["padded"] = {k: res["padded"][k] for k in ("value", "ms_per_step") if k in res["padded"]}
    if isinstance(res.get("full_length"), dict):
        line["full_length"] = {k: res["full_length"][k] for k in ("value", "ms_per_step") if k in res["full_length"]}
    if isinstance(res.get("fwd_bwd_without_optimizer"), dict):
        line["fwd_bwd_without_optimizer"] = res["fwd_bwd_without_optimizer"]
    if isinstance(res.get("secondary"), dict):
        sec = {}
        for k, v in res["secondary"].items():
            if "error" in v:
                sec[k] = {"error": str(v["error"])[:80]}
                continue
            rf = v.get("roofline") or {}
            sec[k] = {"value": v.get("value"), "ms_per_step": v.get("ms_per_step"), "dtype": v.get("dtype_short", v.get("dtype")),
                      "roofline_frac": rf.get("frac"), "roofline_kernel": rf.get("kernel")}
        line["secondary"] = sec
    line["detail"] = "bench_detail.json (also on stderr)"
    # shrink until it fits: the contract's keys, roofline and cpu_baseline are never dropped
    for drop in ("fwd_bwd_without_optimizer", "roofline_fp32_pipe", "full_length", "padded", "parity"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) >= LINE_LIMIT and "secondary" in line:
        line["secondary"] = {k: {"value": v.get("value")} for k, v in line["secondary"].items()}
    if len(json.dumps(line)) >= LINE_LIMIT:
        line.pop("secondary", None)
    for k_, n_ in (("sample", 160), ("workload", 200)):
        tgt = line.get("cpu_baseline", {}) if k_ == "sample" else line.get("config", {})
        if len(json.dumps(line)) >= LINE_LIMIT and isinstance(tgt.get(k_), str):
            tgt[k_] = tgt[k_][:n_]
    assert len(json.dumps(line)) < LINE_LIMIT, "bench line over the limit even after dropping every optional part"
    return line


def emit(res, out=None, err=None, detail_path=None):
    """Full result -> bench_detail.json next to bench.py (best effort) and stderr; THEN the compact line as the last thing on
    stdout.  Nothing is printed to stdout after it."""
    out, err = out or sys.stdout, err or sys.stderr
    full = json.dumps(res)
    try:
        with open(detail_path or os.path.join(ROOT, "bench_detail.json"), "w") as f:
            f.write(full + "\n")
    except OSError as e:  # (a read-only checkout must not cost the line)
        print(f"[bench] bench_detail.json not written: {e}", file=err, flush=True)
    print("[bench detail] " + full, file=err, flush=True)
    line = json.dumps(compact_line(res))
    print(line, file=out, flush=True)
    return line


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def launch_ranks(n, argv, script=None, timeout_s=3000.0, extra_env=None):
    """`bench.py --gpus N` started WITHOUT a launcher (WORLD_SIZE unset): start the N ranks as fresh child processes --
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, as torch.distributed.run would -- wait for
    them, forward rank 0's JSON line, and fail if any rank fails (the others are then stopped by PID: they would wait in
    the rendezvous or a collective forever).  The parent never touches the GPU, and nothing is exec'ed from a process that
    has.  -> (return code, rank 0's stdout)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = script or os.path.abspath(__file__)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # rank 0's stdout is drained by a thread (a full pipe must not block it); the ranks are polled so that one failure
    # ends the job instead of hanging it
    import threading
    out0 = []
    th = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    th.start()
    t_end = time.time() + timeout_s
    rc = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            c = procs[r].poll()
            if c is not None:
                alive.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    log(f"rank {r} exited with code {c}: stopping the other ranks")
        if rc != 0 or time.time() > t_end:
            if rc == 0:
                rc = 124
                log(f"ranks still running after {timeout_s:.0f} s: stopping them")
            for r in alive:
                procs[r].terminate()
            for r in alive:
                try:
                    procs[r].wait(10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            alive = set()
        else:
            time.sleep(0.05)
    th.join(10)
    return rc, (out0[0] if out0 else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--aux", type=int, default=8, help="aux crops: prefix slots = 4*(1+aux)")
    ap.add_argument("--full-length", action="store_true", help="all sequences at full length (worst case)")
    ap.add_argument("--model", default="bert", choices=["bert", "roberta"], help="encoder architecture (base size)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="GEMM arithmetic: fp32 (BASELINE config 2, default) or bf16 compute with fp32 accumulation (configs 3-4)")
    ap.add_argument("--f32-pipe", action="store_true",
                    help="fp32 mode: form the products on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32) instead of the library default, "
                         "six exact bf16 partial products of three-way split fp32 operands (csrc/gemm_f32x3.hip)")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--optimizer", default="all", choices=["all", "reference", "torch"],
                    help="all: mtvaf_amd.optim.AdamW (HIP kernels, per-layer updates enqueued inside the backward pass) over "
                         "every parameter; reference: the same optimizer on the three name-matched groups + linear warm-up "
                         "schedule of modules/train.py:894-926; torch: torch.optim.AdamW(fused=True) after the backward")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward + backward as ONE HIP graph (mtvaf_amd.graph.GraphedTrainStep: device-side dropout "
                         "epoch, eager optimizer step) -- for the launch-bound shapes (bs 4 / S 64, bf16 at bs 32)")
    ap.add_argument("--no-overlap-optimizer", action="store_true", help="HIP AdamW launched by step() only (after the backward)")
    ap.add_argument("--unpad", action="store_true",
                    help="(the default since round 5; kept for old command lines) padding-free execution "
                         "(mtvaf_amd.engine.UNPAD): the encoder layers run on the packed unmasked token rows; loss / tags / "
                         "gradients are those of the padded run")
    ap.add_argument("--padded", action="store_true",
                    help="time the padded run (MTVAF_UNPAD=0: every [B, S] token row computed, the reference's layout) instead of "
                         "the default padding-free one; without the flag the padded rate is reported beside `value` (key `padded`)")
    ap.add_argument("--grad-wire", default="auto", choices=["auto", "fp32", "bf16"],
                    help="N > 1: wire format of the gradient exchange (mtvaf_amd.parallel.GradSync): fp32 = RCCL all_reduce(AVG) in "
                         "place; bf16 = pack -> all_to_all -> fp32 sum -> all_gather (mesh-shaped, half the bytes); auto = fp32 in "
                         "fp32 compute mode, bf16 in bf16 compute mode")
    ap.add_argument("--grad-buckets", type=int, default=4,
                    help="N > 1, bf16 wire: encoder layers are exchanged in this many all_to_all + all_gather pairs per step (4 = "
                         "three BERT-base layers, 85 MB, per exchange); 0 = one exchange per layer")
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1: every rank draws its own ragged batch (seed 1234 + rank) instead of taking its length-balanced share "
                         "of ONE global batch of N x batch sentences (mtvaf_amd.parallel.balanced_shards: sort by length, deal in "
                         "snake order) -- under padding-free execution a rank's step time follows its token rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (fwd+bwd without the optimizer, padding-free): profiler runs, so that "
                         "every kernel launch in the trace belongs to the headline workload")
    a = ap.parse_args()

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # started without a launcher: be the launcher (no GPU call has been made in this process)
        rc, out = launch_ranks(a.gpus, sys.argv[1:])
        line = [ln for ln in out.splitlines() if ln.startswith("{")]
        if rc == 0 and line:
            print(line[-1], flush=True)
            return
        raise SystemExit(rc or 1)
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher started {world} ranks; use "
                         f"--nproc-per-node {a.gpus}, or start `python bench.py --gpus {a.gpus}` without a launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # Launcher smoke test on a 1-GPU box: MTVAF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 over gloo (RCCL refuses
    # two ranks per device).  It exercises rendezvous, GradSync, barriers and the rank-0 JSON line -- not a measurement.
    one_dev = os.environ.get("MTVAF_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))
    from mtvaf_amd import hip
    hip.lib()
    hip.set_compute_dtype(a.dtype)
    if a.f32_pipe:
        hip.f32_split(False)
    split_mode = a.dtype == "fp32" and hip.f32_split()  # (the library default; MTVAF_F32_SPLIT=0 / --f32-pipe: the fp32 MFMA pipe)
    peak_key = "fp32x3" if split_mode else a.dtype

    B, S, P = a.batch, a.seq, 4 * (1 + a.aux)
    from mtvaf_amd import engine
    if a.padded and a.unpad:
        raise SystemExit("--padded and --unpad exclude each other")
    a.unpad = not a.padded and not a.graph  # (a captured step runs the padded layout: the packed row count cannot reach the host)
    engine.UNPAD = bool(a.unpad)
    model, cfg = build_model(device, a.model, S)
    model.train()
    sync = None
    if world > 1:
        from mtvaf_amd.parallel import GradSync
        wire = {"auto": ("bf16" if a.dtype == "bf16" else None), "fp32": None, "bf16": "bf16"}[a.grad_wire]
        # (the bucket count belongs to the bf16 wire; the fp32 wire exchanges one flat buffer per layer in place)
        sync = GradSync(model, compress=wire, layer_buckets=(a.grad_buckets if (a.grad_buckets > 0 and wire == "bf16") else None))
    sched = None
    if a.no_optimizer:
        opt = None
    elif a.optimizer == "reference":
        from mtvaf_amd.optim import build_optimizer
        opt, sched = build_optimizer(model, types.SimpleNamespace(lr=3e-5, warmup_ratio=0.01, use_prefix=True,
                                                                  grad_sync=sync), 100000)
    elif a.optimizer == "torch":
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, fused=True)
    else:
        from mtvaf_amd.optim import AdamW
        opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model,
                    overlap=not (a.graph or a.no_overlap_optimizer), grad_sync=sync)  # (a captured backward cannot carry the per-step learning rate)
    sharding = None
    if world > 1 and not a.no_balance:
        # ONE global batch (the same on every rank: seed 1234), dealt by length so that the ranks' token-row counts agree
        from mtvaf_amd.parallel import balanced_shards
        gb = synthetic_batch(world * B, S, a.aux, cfg.vocab_size, 1234, "cpu", a.full_length)
        mine = torch.tensor(balanced_shards(gb[1].sum(1).tolist(), world)[rank])
        batch = tuple(t[mine].to(device) for t in gb)
        sharding = "length-balanced shares of one global batch (sort by length, snake deal)"
    else:
        batch = synthetic_batch(B, S, a.aux, cfg.vocab_size, 1234 + rank, device, a.full_length)
        if world > 1:
            sharding = "independent per-rank batches (seed 1234 + rank)"
    ids, mask, tt, labels, feats, aux = batch

    def step():
        out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats,
                    aux_imgs=aux)
        out.loss.backward()
        if opt is not None:
            opt.step()
            if sched is not None:
                sched.step()
            opt.zero_grad(set_to_none=True)
        else:
            for p in model.parameters():
                p.grad = None
        assert len(out.logits) == B  # the trainer reads the decoded tags after the update (modules/train.py:627-647)
        return out

    eager_step = step
    if a.graph:
        if world > 1:
            raise SystemExit("--graph is a single-GPU option (the gradient all-reduce hooks are not captured)")
        from mtvaf_amd.graph import GraphedTrainStep
        gstep = GraphedTrainStep(model, dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels,
                                             imagelabel=None, images=feats, aux_imgs=aux))

        def step():  # noqa: F811
            out = gstep(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
            if opt is not None:
                opt.step()
                if sched is not None:
                    sched.step()
                opt.zero_grad(set_to_none=True)
            assert len(out.logits) == B
            return out

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log("model and batch built")
    import gc
    for i in range(a.warmup):
        if i == a.warmup - 1:
            # set-up garbage (model construction, the first steps' graphs) is collected BEFORE the last warm-up step, not inside a
            # timed step -- and not between the warm-up and the timed region either: the collection idles the GPU for ~0.1 s, its
            # clocks drop, and the first timed step then read 11.2 ms against 8.9 - 9.0 for the other 39 (profiles/r06_step_ms.txt)
            torch.cuda.synchronize()
            gc.collect()
        step()
    if a.warmup == 0:
        gc.collect()
    if rank == 0:
        log(f"{a.warmup} warm-up steps enqueued")
    barrier()
    # per-step HIP events on the main stream (no host sync inside the timed region): the median step is reported next to
    # the bracketed wall-clock figure, which stays `value` (the driver's contract)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        out = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0  # this rank's own K steps (before it waits for the others)
    barrier()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    med_ms = statistics.median(step_ms)
    rank_ms_spread = None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        # per-rank step times (each rank's batch is ragged in its own way): (max - min) / mean over the ranks
        tl = torch.zeros(world, device=device, dtype=torch.float64)
        tl[rank] = dt_local
        dist.all_reduce(tl)
        rank_ms_spread = round(float((tl.max() - tl.min()) / tl.mean()), 4)
    loss_val = float(out.loss.detach())
    if hip.streamk_errors():
        raise SystemExit("a stream-K launch of the bf16 GEMM reported a timed-out wait (mtvaf_amd.hip.streamk_errors)")
    log(f"timed region done: {dt:.3f}s for {a.steps} steps, loss {loss_val:.4f}")
    value = world * B * a.steps / dt
    per_gpu = value / world
    ftrain = 3 * f_fwd(S, P)

    fwd_bwd_only = None
    if opt is not None and not a.graph and not a.no_secondary:
        # secondary figure: the same K steps without the optimizer update (the metric's literal "fwd+bwd")
        def step_nb():
            out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None,
                        images=feats, aux_imgs=aux)
            out.loss.backward()
            for p_ in model.parameters():
                p_.grad = None
            assert len(out.logits) == B
        if hasattr(opt, "suspended"):
            opt.suspended = True  # no step() follows these backward passes
        step_nb()
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step_nb()
        barrier()
        dt1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt1], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt1 = float(t)
        fwd_bwd_only = {"value": round(world * B * a.steps / dt1, 2), "ms_per_step": round(1e3 * dt1 / a.steps, 3)}
        if hasattr(opt, "suspended"):
            opt.suspended = False

    # flops of the rows that are real tokens (what a padding-free run executes): per sentence with its own length
    lens_host = mask.sum(1).tolist()
    f_exec = 3 * sum(f_fwd(int(n), P) for n in lens_host) / B
    real_rows = sum(lens_host) / float(B * S)
    padded = None
    if a.unpad and not a.no_secondary and real_rows < 0.97:
        # secondary figure: the SAME K steps (optimizer included) in the padded layout (MTVAF_UNPAD=0): every [B, S] token row
        # computed, as the reference does
        engine.UNPAD = False
        try:
            for _ in range(2):
                step()
            barrier()
            t2 = time.perf_counter()
            for _ in range(a.steps):
                step()
            barrier()
            dt2 = time.perf_counter() - t2
        finally:
            engine.UNPAD = True
        if world > 1:
            t = torch.tensor([dt2], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t)
        v2 = world * B * a.steps / dt2
        padded = {"value": round(v2, 2), "ms_per_step": round(1e3 * dt2 / a.steps, 3),
                  "mfma_fraction_of_step_algorithmic": round(v2 / world * ftrain / (PEAK_TFLOPS[peak_key] * 1e12), 4),
                  "note": "same workload and results (loss, tags, parameter gradients: tests/test_unpad_gpu.py, pad_mode fixture) with "
                          "every masked token row computed (`--padded` / MTVAF_UNPAD=0); its fraction credits SURVEY 8d's F_train"}

    full_length = None
    if not a.full_length and not a.graph and not a.no_secondary and world == 1:
        # secondary figure: the same K steps on a batch whose sequences are ALL S tokens long -- nothing is masked, so neither the
        # k-tile lists of the weight gradients nor padding-free execution have anything to skip: the floor of both
        ragged = (ids, mask, tt, labels, feats, aux)
        ids, mask, tt, labels, feats, aux = synthetic_batch(B, S, a.aux, cfg.vocab_size, 1234 + rank, device, True)
        try:
            for _ in range(2):
                step()
            barrier()
            t3 = time.perf_counter()
            for _ in range(a.steps):
                step()
            barrier()
            dt3 = time.perf_counter() - t3
        finally:
            ids, mask, tt, labels, feats, aux = ragged
        full_length = {"value": round(B * a.steps / dt3, 2), "ms_per_step": round(1e3 * dt3 / a.steps, 3)}

    res = {"metric": "training sentences/sec (fwd+bwd)", "value": round(value, 2), "unit": "sentences/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": ("fp32 (3xbf16 split products, fp32 accumulate)" if split_mode else "fp32 (fp32 MFMA pipe)") if a.dtype == "fp32"
                    else "bf16 MFMA / fp32 accumulate+storage",
           "data": "synthetic",
           "config": {"workload": f"TVNetSAModel2 {'RoBERTa' if a.model == 'roberta' else 'BERT'}-base random-init, fwd+bwd{' as one HIP graph' if a.graph else ''}{'' if a.no_optimizer else {'torch': '+AdamW(torch fused)'}.get(a.optimizer, '+AdamW(HIP' + (', eager after the replay)' if a.graph else ', overlapped with backward)'))}, "
                                  f"bs={B}/GPU, seq_len={S}, {P} visual prefix slots (1+{a.aux} region-feature "
                                  f"images through the prompt generator), train mode (dropout live), "
                                  f"{'full-length' if a.full_length else 'ragged 16..S'} sequences"
                                  f"{', padding-free execution (masked token rows not computed)' if a.unpad else ', padded execution (every [B, S] row computed)'}"
                                  f"{', encoder GEMM operands as pre-split plane images' if (a.unpad and split_mode and _planes_on()) else ''}",
                      "global_batch": B * world, "seq_len": S, "prefix": P,
                      "parallelism": f"dp{world}" + (" (RCCL all-reduce overlapped with backward)" if world > 1 else "")},
           "median_ms_per_step": round(med_ms, 3), "value_median": round(world * B / (med_ms * 1e-3), 2),
           "step_ms": [round(x, 3) for x in step_ms],  # (detail file only: HIP events on the main stream around every timed step)
           "loss": round(loss_val, 4),
           # padding-free: the flops of the rows that are real tokens (per sentence with its own length); padded: SURVEY 8d's F_train
           "mfma_fraction_of_step": round(per_gpu * (f_exec if a.unpad else ftrain) / (PEAK_TFLOPS[peak_key] * 1e12), 4),
           "peak_tflops": PEAK_TFLOPS[peak_key],
           "flop_per_sentence_train": ftrain, "fwd_bwd_without_optimizer": fwd_bwd_only,
           "real_token_rows": round(real_rows, 4), "flop_per_sentence_train_real_rows": round(f_exec),
           "note_flops": "padding-free run: mfma_fraction_of_step = sentences/s x 3 x F_fwd(len_b, P) averaged over the batch's OWN sentence "
                         "lengths (the flops of real token rows) / peak_tflops; padded run (--padded): the algorithmic 3 x F_fwd(S, P) "
                         "of SURVEY 8d.  peak_tflops in split mode: 2500 / 6 = 416.7 TFLOP/s of fp32-equivalent work on the bf16 pipe.  "
                         "The roofline object counts the flops its launches execute (packed rows; k-tile lists)",
           "padding": "skipped (default): every fraction counts EXECUTED flops" if a.unpad else "computed (reference behaviour)",
           "padded": padded, "full_length": full_length}
    res["rank_ms_spread"] = rank_ms_spread
    if world > 1:
        rows = torch.zeros(world, device=device, dtype=torch.float64)
        rows[rank] = float(mask.sum())
        dist.all_reduce(rows)
        res["rank_token_rows"] = [int(rows.min()), int(rows.max())]  # unmasked token rows per rank: what a padding-free step computes
        res["sharding"] = sharding
        try:
            res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
        except Exception:
            res["rccl_version"] = None
    if rank_ms_spread is not None and rank_ms_spread >= 0.03:
        log(f"per-rank step times differ by {100 * rank_ms_spread:.1f} % (>= 3 %): the slowest rank sets `value`")
    res["n_ranks_seen"] = dist.get_world_size() if world > 1 else 1
    res["backend"] = dist.get_backend() if world > 1 else None
    if sync is not None:
        # 3 further steps with an event pair around every exchange (outside the timed region: the events are not free)
        sync.timing = True
        for _ in range(3):
            step()
        tm = sync.take_timing()
        sync.timing = False
        res["grad_sync"] = {"wire": sync.compress or "fp32", "wire_fallback": "--grad-wire fp32 (RCCL all_reduce(AVG) in place)",
                            "layer_exchanges_per_step": sync.layer_buckets or len(sync.encoder.layer),
                            "comm_stream_ms_per_step": round(tm["comm_stream_ms"] / max(1, tm["passes"]), 3),
                            "exposed_tail_ms_per_step": round(tm["exposed_tail_ms"] / max(1, tm["passes"]), 3),
                            "note": "rank 0, 3 extra steps: time the communication stream spent in exchanges (and the per-layer "
                                    "optimizer updates queued behind them), and how long the last exchange ran past the "
                                    "backward pass's own kernels"}

    # ---- roofline of the dominant kernel (fp32 MFMA GEMM), measured live with HIP events recorded by the
    # library on the launch stream, directly around each main GEMM kernel of 3 further identical steps ----
    if rank == 0 and not a.no_roofline:
        if sync is not None:
            sync.enabled = False
        if a.graph:
            gstep.close()  # the profiled steps run eagerly (the launch profiler brackets individual launches)
        res["roofline"] = roofline_pass(eager_step, mask, B, S, a.dtype, a.unpad, peak_key="fp32x3" if split_mode else None)
        ex = res["roofline"]["all_gemm_kernels"]["executed_tflop_per_step"]
        # the whole step by the flops its GEMM launches EXECUTE (the weight-gradient products skip the k-tiles of masked
        # token rows) plus the attention products' algorithmic share, next to the algorithmic figure above (split mode: both
        # against the fp32-equivalent peak of the bf16 pipe; the attention products still run the fp32 pipe)
        attn_tflop = (sum(3 * 12 * 4 * int(n) * (int(n) + P) * 768 for n in lens_host) if a.unpad
                      else B * 3 * 12 * 4 * S * (S + P) * 768) / 1e12
        res["mfma_fraction_of_step_executed"] = round((ex + attn_tflop) / (1e-3 * res["ms_per_step"]) / PEAK_TFLOPS[peak_key], 4)
    if rank == 0:
        log("roofline pass done")
    if rank == 0 and world == 1 and not a.no_secondary and not a.graph and (B, S, a.aux, a.dtype, a.model, a.unpad) == (32, 128, 8, "fp32", "bert", True):
        # the other BASELINE configurations that fit one GPU, measured by the same process AFTER the headline (secondary
        # figures: the headline's value / config / dtype stay those of configs[1])
        del model, opt
        step = eager_step = None  # noqa: F841
        torch.cuda.empty_cache()
        res["secondary"] = {}
        for key, (dt_, arch_, b_, s_, aux_) in {"c1_fp32": ("fp32", "bert", 4, 64, 3), "c1_fp32_graph": ("fp32", "bert", 4, 64, 3),
                                                ("c2_fp32_pipe" if split_mode else "c2_fp32_split"): ("fp32", "bert", 32, 128, 8),
                                                "c3_bf16": ("bf16", "roberta", 32, 128, 8), "c4_bf16": ("bf16", "bert", 64, 128, 8),
                                                # configs[4], the roofline stress shape (65 536 token rows before packing): five
                                                # training steps each, in both compute modes
                                                "c5_bf16": ("bf16", "bert", 128, 512, 8), "c5_fp32": ("fp32", "bert", 128, 512, 8)}.items():
            try:
                # (C1 is the one host-bound configuration: 10 steps behind 3 warm-up steps read 750-930 sentences/s from call
                # to call, 40 behind 10 read what `bench.py --batch 4 --seq 64 --aux 3` reads)
                res["secondary"][key] = secondary_config(key, device, dt_, arch_, b_, s_, aux_,
                                                         split=("_split" in key) or (bool(split_mode) and "_pipe" not in key),
                                                         unpad=not key.endswith("_graph"), graph=key.endswith("_graph"),
                                                         steps=40 if key.startswith("c1_") else (5 if key.startswith("c5_") else 10),
                                                         warmup=10 if key.startswith("c1_") else (2 if key.startswith("c5_") else 5))
            except Exception as e:  # a secondary figure must never cost the headline line
                res["secondary"][key] = {"error": repr(e)}
            torch.cuda.empty_cache()
        # the same workload as `value` in the other fp32 arithmetic (kept in the line for continuity with rounds 1-3, whose
        # headline was the fp32 MFMA pipe)
        for key_, tag_ in (("c2_fp32_pipe", "fp32_pipe"), ("c2_fp32_split", "fp32_split")):
            sp = res["secondary"].get(key_, {})
            if "value" in sp:
                res["value_" + tag_] = sp["value"]
                res["ms_per_step_" + tag_] = sp["ms_per_step"]
                res["roofline_" + tag_] = sp.get("roofline")
        log("secondary configurations done")
    if rank == 0 and world == 1 and not a.no_secondary and not a.no_cpu_baseline:
        try:  # (detail only; must never cost the line)
            res["frontend_cache_build"] = frontend_cache_throughput(device)
            res["frontend_cache_build_bf16"] = frontend_cache_throughput(device, compute="bf16")
        except Exception as e:
            res["frontend_cache_build"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(B, S, a.aux)
        if (B, S, a.aux, a.dtype, a.model) == (32, 128, 8, "fp32", "bert") and not a.graph:
            # measured deviation of the assembled HIP path from the oracle at this very shape, in the arithmetic and layout that
            # were timed (one more oracle step on the host: the checker, after the timed region)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import parity_report
                res["parity"] = parity_report.report(B, S, a.aux, device=device)
                log(parity_report.fmt(res["parity"]))
            except Exception as e:  # (must never cost the line)
                res["parity"] = {"error": repr(e)}
        if (B, S, a.aux) == (32, 128, 8):  # second entry: the reference's own CPU-runnable configuration (configs[0])
            res["cpu_baseline_c1"] = cpu_baseline(4, 64, 3, seconds_budget=10.0)
        log("cpu baseline done")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(res)


if __name__ == "__main__":
    main()
