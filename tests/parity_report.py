"""Measured deviation of the ASSEMBLED HIP path from the CPU oracle at a BASELINE shape -- the figures behind the parity tests'
tolerances, made visible: `__graft_entry__.smoke()` prints them for the bench shape (C2: bs 32, S 128, P 36),
`bench.py`'s cpu_baseline leg puts them into the JSON line (`parity`), and tests/test_configs_gpu.py asserts on the same numbers.

Test infrastructure: imports `oracle/` (the checker); never imported by anything under `mtvaf_amd/`.

    emissions_max_rel        max |em - oracle| / max |oracle|            (the max-norm figure `close()` bounds)
    emissions_elem_rel_p99   99th percentile / maximum of |em - oracle| / |oracle| over the elements with |oracle| >= FLOOR x max
    emissions_elem_rel_max   |oracle| (element-wise relative error; FLOOR = 1e-2: 99 % of the emissions at the bench shape)
    loss_rel                 |loss - oracle| / |oracle|
    tags_equal               decoded Viterbi paths == the oracle's (bit-exact index work)
    worst_grad_rel           max over six parameter gradients (prompt generator, projector, two encoder weights, fc, CRF) of
                             max |g - oracle| / max |oracle|, and its name
"""
from __future__ import annotations

import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FLOOR = 1e-2
GRADS = ["encoder_conv.2.weight", "projectors.0.weight", "bert.encoder.layer.5.intermediate.dense.weight",
         "bert.encoder.layer.0.attention.self.value.weight", "fc.weight", "crf.transitions"]


def elementwise_rel(got: torch.Tensor, ref: torch.Tensor, floor: float = FLOOR):
    """-> (p99, max, share of elements above the floor) of |got - ref| / |ref| over the elements with |ref| >= floor * max |ref|."""
    got, ref = got.detach().float().cpu().reshape(-1), ref.detach().float().cpu().reshape(-1)
    keep = ref.abs() >= floor * float(ref.abs().max())
    rel = ((got - ref).abs() / ref.abs())[keep]
    if rel.numel() == 0:
        return 0.0, 0.0, 0.0
    return float(torch.quantile(rel.double(), 0.99)), float(rel.max()), float(keep.float().mean())


def deviations(em, oem, loss, oloss, tags, otags, grads, ograds, valid=None):
    """The report dict from the two runs' results (emissions restricted to `valid` [B,S] bool when given: padding-free execution
    does not compute masked positions)."""
    em, oem = em.detach().float().cpu(), oem.detach().float().cpu()
    loss = float(loss.detach()) if torch.is_tensor(loss) else float(loss)
    oloss = float(oloss.detach()) if torch.is_tensor(oloss) else float(oloss)
    if valid is not None:
        em, oem = em[valid], oem[valid]
    p99, emax, share = elementwise_rel(em, oem)
    worst, wname = 0.0, ""
    for n in ograds:
        g, og = grads[n].detach().float().cpu(), ograds[n].detach().float().cpu()
        r = float((g - og).abs().max() / og.abs().max())
        if r >= worst:
            worst, wname = r, n
    mism = sum(a != b for ta, tb in zip(tags, otags) for a, b in zip(ta, tb)) + sum(len(a) != len(b) for a, b in zip(tags, otags))
    return {"emissions_max_rel": float((em - oem).abs().max() / oem.abs().max()),
            "emissions_elem_rel_p99": p99, "emissions_elem_rel_max": emax, "elem_floor": FLOOR, "elem_share": round(share, 4),
            "loss_rel": abs(loss - oloss) / abs(oloss), "tags_equal": mism == 0, "tag_mismatches": int(mism),
            "worst_grad_rel": worst, "worst_grad": wname}


def report(B=32, S=128, n_aux=8, seed=41, device="cuda", threads=None):
    """One forward + backward of the assembled `TVNetSAModel2` (eval mode: dropout off; the library's default arithmetic and
    layout) and of the oracle on the same seeded weights and inputs -> the deviations + how long the oracle step took."""
    import params as P
    from oracle import mtvaf_oracle as O
    import test_model_gpu as T
    from mtvaf_amd import engine
    cfg = P.BASE_BERT
    if threads:
        torch.set_num_threads(threads)
    sde, sdh, sdp = P.encoder_params(cfg, seed, std=0.03), P.head_params(cfg, seed + 1), P.prompt_params(seed + 2)
    ids, mask, tt, labels = P.text_batch(cfg, seed + 3, B, S, lo_id=1000)
    labels[:, 0] = 9
    feats, aux, lab = T._prompt_inputs(seed + 4, B, n_aux)
    # oracle (reference models/bert_model.py:480-588 restated: oracle/mtvaf_oracle.py)
    sd = {**{"bert." + k: v.clone() for k, v in sde.items()}, **{k: v.clone() for k, v in sdh.items()},
          **{k: v.clone() for k, v in sdp.items()}}
    for n in GRADS:
        sd[n].requires_grad_(True)
    t0 = time.perf_counter()
    res, _, _ = O.visual_prompt(sd, feats.reshape(B, 4, -1), [aux[:, i].reshape(B, 4, -1) for i in range(n_aux)],
                                num_layers=cfg.layers, num_heads=cfg.heads)
    oloss, oem, otags, _ = O.tvnet2_forward(sd, ids, mask, tt, labels, res, cfg.layers, cfg.heads, cfg.eps)
    oloss.backward()
    t_oracle = time.perf_counter() - t0
    ograds = {n: sd[n].grad for n in GRADS}
    # HIP path, through the module interface the reference trainer calls
    m = T.build_tvnet2(cfg, T.make_args(alpha=0.0, device=device), sde=sde, sdh=sdh, sdp=sdp).eval()
    cap = {}
    decode = m.crf.decode_deferred

    def spy(e, mk):
        cap["em"] = e.detach().clone()
        return decode(e, mk)
    m.crf.decode_deferred = spy
    try:
        out = m(input_ids=ids.to(device), attention_mask=mask.to(device), token_type_ids=tt.to(device), labels=labels.to(device),
                imagelabel=lab.to(device), images=feats.to(device), aux_imgs=aux.to(device))
    finally:
        del m.crf.decode_deferred
    out.loss.backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    rep = deviations(cap["em"], oem, out.loss, oloss, list(out.logits), otags, {n: named[n].grad for n in GRADS}, ograds,
                     valid=mask.bool() if engine.LAST_PACK is not None else None)
    rep.update(shape=f"bs{B}/S{S}/P{4 * (1 + n_aux)}", layout="padding-free" if engine.LAST_PACK is not None else "padded",
               oracle_step_s=round(t_oracle, 3))
    return rep


def fmt(rep) -> str:
    return (f"parity {rep.get('shape', '')} ({rep.get('layout', '')}): emissions max-rel {rep['emissions_max_rel']:.2e}, element-wise rel "
            f"p99 {rep['emissions_elem_rel_p99']:.2e} / max {rep['emissions_elem_rel_max']:.2e} (|ref| >= {rep['elem_floor']:g} max, "
            f"{100 * rep['elem_share']:.1f} % of the elements), loss rel {rep['loss_rel']:.2e}, tags "
            f"{'bit-exact' if rep['tags_equal'] else str(rep['tag_mismatches']) + ' mismatches'}, worst gradient rel "
            f"{rep['worst_grad_rel']:.2e} ({rep['worst_grad']})")
