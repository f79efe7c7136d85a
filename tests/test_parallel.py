"""Data-parallel gradient sync (mtvaf_amd.parallel.GradSync).

CPU part: world_size-2 gloo processes drive the hook protocol (layer_done from inside a backward pass, the
end-of-backward callback, fast flat-buffer path and the accumulate fallback) with a stand-in encoder, and
check the result against the mean of the per-rank gradients.  GPU part: the real model with a 1-rank RCCL
group and force=True exercises the stream/event choreography on the MI355X."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeLayer(torch.nn.Module):
    def __init__(self, n):
        super().__init__()
        self.intermediate = torch.nn.Module()
        self.intermediate.dense = torch.nn.Linear(n, n)
        self.extra = torch.nn.Parameter(torch.zeros(n))

    def ordered_params(self):
        return [self.intermediate.dense.weight, self.intermediate.dense.bias, self.extra]


class _FakeStore:
    def __init__(self, layer):
        ps = layer.ordered_params()
        self.grad = torch.zeros(sum(p.numel() for p in ps))
        self.views, off = [], 0
        for p in ps:
            self.views.append((off, p.shape))
            off += p.numel()

    def grad_views(self):
        return [self.grad[o:o + s.numel()].view(s) for o, s in self.views]


class _FakeEncoder(torch.nn.Module):
    """Mimics BertEncoder's contract with GradSync: per-layer flat gradient buffers handed out through a
    GradSink, layer_done fired newest layer first from inside backward."""

    def __init__(self, L=3, n=8):
        super().__init__()
        from mtvaf_amd.models.modeling_bert import GradSink
        self.layer = torch.nn.ModuleList([_FakeLayer(n) for _ in range(L)])
        self._stores = [_FakeStore(l) for l in self.layer]
        self._sink = GradSink(self._stores)

    @property
    def grad_sink(self):
        return self._sink

    def forward(self, x):
        enc = self

        class F(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, *params):
                ctx.save_for_backward(x)
                return x.sum() * sum((p * (i + 1)).sum() for i, p in enumerate(params))

            @staticmethod
            def backward(ctx, g):
                (x,) = ctx.saved_tensors
                params = [p for l in enc.layer for p in l.ordered_params()]
                views = enc._sink.acquire(params)
                out = [None] * len(params)
                idx = 0
                per_layer = []
                for li, l in enumerate(enc.layer):
                    n = len(l.ordered_params())
                    per_layer.append((idx, n))
                    idx += n
                for li in range(len(enc.layer) - 1, -1, -1):
                    s, n = per_layer[li]
                    for j in range(s, s + n):
                        val = torch.full_like(params[j], float(x.sum()) * (j + 1)) * g
                        if views is not None:
                            views[j].copy_(val)
                            out[j] = views[j]
                        else:
                            out[j] = val
                    enc._sink.layer_done(li)
                return (None, *out)

        params = [p for l in self.layer for p in l.ordered_params()]
        return F.apply(x, *params)


class _FakeModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = _FakeEncoder()
        self.head = torch.nn.Linear(4, 1)
        self.table = torch.nn.Parameter(torch.ones(64, 8))  # "large" parameter: reduced early by its own hook

    def forward(self, x):
        return self.encoder(x) + self.head(x[:4]).sum() + (self.table * x).sum()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mtvaf_amd.parallel import GradSync
    torch.manual_seed(0)
    m = _FakeModel()
    sync = GradSync(m, big_numel=256)
    assert len(sync._early_done) == 0
    x = torch.arange(8, dtype=torch.float32) * (rank + 1)
    res = {}
    # fast path: .grad is None -> flat buffers adopted and all-reduced per layer
    m(x).backward()
    res["fast"] = [p.grad.tolist() for p in m.parameters()]
    aliased = m.encoder.layer[1].intermediate.dense.weight.grad.data_ptr() == m.encoder._stores[1].grad.data_ptr()
    res["aliased"] = aliased
    # accumulate path: second backward without zero_grad
    m(x).backward()
    res["acc"] = [p.grad.tolist() for p in m.parameters()]
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process oracle: mean over ranks of the local gradients
    torch.manual_seed(0)
    ref_m = _FakeModel()
    locals_ = []
    for r in range(world):
        ref_m.zero_grad(set_to_none=True)
        ref_m(torch.arange(8, dtype=torch.float32) * (r + 1)).backward()
        locals_.append([p.grad.clone() for p in ref_m.parameters()])
    mean = [sum(g) / world for g in zip(*locals_)]
    for r in range(world):
        assert out[r]["aliased"], "fast path must adopt the flat layer buffer without a copy"
        for got, want in zip(out[r]["fast"], mean):
            torch.testing.assert_close(torch.tensor(got), want)
        # second backward accumulates: local grad added to the (already averaged) first one, then the
        # accumulated tensors are averaged again by the fallback path -> (mean + local)/.. differs per rank;
        # the invariant that must hold is equality ACROSS ranks after the sync
    for a, b in zip(out[0]["acc"], out[1]["acc"]):
        torch.testing.assert_close(torch.tensor(a), torch.tensor(b))


@pytest.mark.gpu
def test_gradsync_single_rank_rccl_on_gpu():
    """1-rank RCCL group with force=True: the real encoder's layer hooks, side stream, events and the
    end-of-backward bucket run on the MI355X and must leave the gradients unchanged (mean over 1 rank)."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import params as P
    from transformers import BertConfig
    from mtvaf_amd.models.bert_model import TVNetSAModel2
    from mtvaf_amd.parallel import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        cfg = BertConfig(vocab_size=500, hidden_size=128, num_hidden_layers=3, num_attention_heads=2,
                         intermediate_size=256, max_position_embeddings=64, hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0)
        args = types.SimpleNamespace(bert_name="bert-base-uncased", bert_config=cfg, use_prefix=False, vao=False,
                                     noauxloss=True, use_probe=False, n_gpu=1, alpha=0.0, prefix_len=4, prefix_dim=768,
                                     device="cuda", resnet_root=None, use_152=False)
        labels_list = ["O", "B-NEU", "I-NEU", "B-POS", "I-POS", "B-NEG", "I-NEG", "X", "[CLS]", "[SEP]"]
        torch.manual_seed(0)
        m = TVNetSAModel2(labels_list, None, args).to("cuda").eval()  # head dropout (p = 0.1) off: deterministic
        ids, mask, tt, labels = (t.to("cuda") for t in P.text_batch(P.EncCfg(vocab_size=500), 3, 8, 32, lo_id=5))
        m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels).loss.backward()
        ref = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        m.zero_grad(set_to_none=True)
        sync = GradSync(m, force=True)
        m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels).loss.backward()
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            if p.grad is not None:
                # the word table is scatter-added with float atomics (order-dependent in the last bits)
                tol = dict(rtol=1e-3, atol=1e-5) if "word_embeddings" in n else dict(rtol=1e-5, atol=1e-6)
                torch.testing.assert_close(p.grad, ref[n], msg=n, **tol)
        st = m.bert.encoder._stores[1]
        g = m.bert.encoder.layer[1].intermediate.dense.weight.grad
        assert st.grad.data_ptr() <= g.data_ptr() < st.grad.data_ptr() + st.grad.numel() * 4
    finally:
        dist.destroy_process_group()


def _gpu_worker(rank, world, port, q):
    """Two ranks on ONE MI355X over gloo (RCCL refuses two ranks per device): the real model, the encoder's
    weight-gradient side stream (M = 1024), the per-layer hooks, the early word-table hook and the tail bucket."""
    import types
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import params as P
        from transformers import BertConfig
        from mtvaf_amd.models.bert_model import TVNetSAModel2
        from mtvaf_amd.parallel import GradSync
        torch.cuda.set_device(0)
        cfg = BertConfig(vocab_size=3000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                         intermediate_size=256, max_position_embeddings=64, hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0)
        # with the visual prefix: the prompt generator runs on the second stream (forward AND backward), its
        # encoder_conv weights take the early all-reduce hook from there
        args = types.SimpleNamespace(bert_name="bert-base-uncased", bert_config=cfg, use_prefix=True, vao=False,
                                     noauxloss=True, use_probe=False, n_gpu=1, alpha=0.0, prefix_len=4, prefix_dim=768,
                                     device="cuda", resnet_root=None, use_152=False)
        labels_list = ["O", "B-NEU", "I-NEU", "B-POS", "I-POS", "B-NEG", "I-NEG", "X", "[CLS]", "[SEP]"]
        torch.manual_seed(0)
        m = TVNetSAModel2(labels_list, None, args).to("cuda").eval()
        batches = []
        for r in range(world):
            g = torch.Generator().manual_seed(100 + r)
            feats = torch.randn(16, 3840, 2, 2, generator=g).abs().to("cuda")
            aux = torch.randn(16, 3, 3840, 2, 2, generator=g).abs().to("cuda")
            batches.append(tuple(t.to("cuda") for t in P.text_batch(P.EncCfg(vocab_size=3000), 11 + r, 16, 64, lo_id=5))
                           + (feats, aux))

        def grads(batch):
            ids, mask, tt, labels, feats, aux = batch
            m.zero_grad(set_to_none=True)
            m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux).loss.backward()
            torch.cuda.synchronize()
            return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

        local = [grads(b) for b in batches]                       # no sync installed yet
        want = {n: sum(g[n] for g in local) / world for n in local[0]}
        sync = GradSync(m, big_numel=1 << 16)                      # the 3000x128 word table takes the early hook
        got = grads(batches[rank])
        worst = 0.0
        for n, w in want.items():
            err = float((got[n] - w).abs().max()) / (float(w.abs().max()) + 1e-12)
            worst = max(worst, err)
        aliased = m.bert.encoder._stores[1].grad is not None
        q.put((rank, worst, aliased, len(want)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_gradsync_two_ranks_real_model_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, worst, aliased, n in out:
        assert n >= 40 and aliased
        assert worst < 2e-4, f"rank {rank}: synced gradients differ from the mean of the per-rank gradients by {worst:.2e}"
