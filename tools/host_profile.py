"""cProfile of the host side of training steps (where does the Python time between kernel launches go?)."""
import cProfile, os, pstats, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda"
model, cfg = bench.build_model(dev, "bert", S)
model.train()
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(B, S, 8, cfg.vocab_size, 0, dev)
opt = torch.optim.AdamW(model.parameters(), lr=3e-5, fused=True)
def step():
    out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
    out.loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
    return len(out.logits)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
