"""Frozen visual front-end (row f1): the ResNet trunks match torchvision's published layouts, the pyramid
pooling follows ImageModel.get_resnet_prompt, and cached region features reproduce the raw-image prefix."""
import types

import pytest
import torch


def test_resnet_trunks_have_torchvision_layout():
    from mtvaf_amd.models import resnet as R
    # published torchvision parameter counts
    counts = {"resnet18": 11_689_512, "resnet34": 21_797_672, "resnet50": 25_557_032, "resnet101": 44_549_160,
              "resnet152": 60_192_808}
    for name, n in counts.items():
        m = getattr(R, name)()
        assert sum(p.numel() for p in m.parameters()) == n, name
        assert [k for k, _ in m.named_children()] == ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3",
                                                      "layer4", "avgpool", "fc"]
    sd = R.resnet50().state_dict()
    for k in ("conv1.weight", "bn1.running_mean", "layer1.0.downsample.0.weight", "layer1.0.downsample.1.bias",
              "layer3.5.conv3.weight", "layer4.2.bn3.num_batches_tracked", "fc.bias"):
        assert k in sd, k
    assert tuple(sd["layer2.0.conv2.weight"].shape) == (128, 128, 3, 3) and R.resnet50().layer2[0].conv2.stride == (2, 2)


def test_pyramid_pooling_and_feature_cache_cpu():
    from mtvaf_amd.features import RegionFeatureCache
    from mtvaf_amd.models.bert_model import ImageModel
    torch.manual_seed(0)
    im = ImageModel(use_18=True, resnet_root="random")
    x = torch.randn(2, 3, 64, 64)
    aux = torch.randn(2, 3, 3, 64, 64)
    im.eval()
    pyr, aux_pyr = im(x, aux)
    assert [tuple(p.shape) for p in pyr] == [(2, 64, 2, 2), (2, 128, 2, 2), (2, 256, 2, 2), (2, 512, 2, 2)]
    assert len(aux_pyr) == 3 and tuple(aux_pyr[0][3].shape) == (2, 512, 2, 2)
    # layer1 output is 16x16 at 64x64 input: kernel 8 -> the pooled cell is the mean of an 8x8 quadrant
    h = im.resnet.layer1(im.resnet.maxpool(im.resnet.relu(im.resnet.bn1(im.resnet.conv1(x)))))
    assert torch.allclose(pyr[0][:, :, 0, 1], h[:, :, :8, 8:].mean((2, 3)), atol=1e-6)
    cache = RegionFeatureCache(im)
    im.train()
    feats, fa = cache.extract(x, aux)
    assert im.training and tuple(feats.shape) == (2, 960, 2, 2) and tuple(fa.shape) == (2, 3, 960, 2, 2)
    assert torch.equal(feats, torch.cat(pyr, 1)) and torch.equal(fa[:, 1], torch.cat(aux_pyr[1], 1))
    cache.add(["a", "b"], x, aux)
    f2, a2 = cache.batch(["b", "a"], "cpu")
    assert torch.equal(f2[0], feats[1]) and torch.equal(a2[1], fa[0])


def _frontend_case(name, device):
    """(drop-in ImageModel with the fixture's seeded trunk, images, aux images, fixture arrays)"""
    import os
    import sys
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    import params as P
    from mtvaf_amd.models.bert_model import ImageModel
    fx = np.load(os.path.join(here, f"frontend_{name}.npz"))
    im = ImageModel(use_18=(name == "resnet18"), resnet_root="random")
    sd = P.resnet_params(im.resnet, int(fx["seed"]))
    im.resnet.load_state_dict(sd)
    x, aux = P.image_batch(int(fx["seed"]) + 1, int(fx["B"]), int(fx["n_aux"]), int(fx["hw"]))
    return im.to(device), sd, x.to(device), aux.to(device), fx


LEVEL_TOL = 2e-3  # per pyramid level, relative error in norm (GPU; and hosts whose oneDNN sums in another order)


def _levels(t, name):
    """[..., C, 2, 2] pyramid -> its four levels (the channel blocks of layer1..layer4 that get_resnet_prompt concatenates,
    bert_model.py:101-111)."""
    widths = (64, 128, 256, 512) if name == "resnet18" else (256, 512, 1024, 2048)
    return torch.split(t, widths, dim=-3)


def _close_per_level(got, ref, name, msg, device, strict):
    """strict = (rtol, atol): element-wise first (it holds on the CPU that wrote the fixture); wherever it does not -- another
    host's oneDNN kernel selection, MIOpen's algorithm choice on the GPU -- EVERY level of the pyramid must still agree to
    LEVEL_TOL in norm (a wrong pooling window or tap on one level moves that level by O(1), which a whole-tensor norm over
    3840 channels would dilute), and no single element may be far off."""
    if strict is not None:
        try:
            torch.testing.assert_close(got, ref, rtol=strict[0], atol=strict[1])
            return "element-wise"
        except AssertionError:
            pass
    worst = 0.0
    for li, (g, r) in enumerate(zip(_levels(got, name), _levels(ref, name))):
        rel = float((g.double() - r.double()).norm() / r.double().norm())
        worst = max(worst, rel)
        assert rel < LEVEL_TOL, (msg, f"level {li + 1}", rel)
    print(f"[frontend {name} {device}] {msg}: per-level relative error in norm <= {worst:.2e}", flush=True)
    if device == "cpu":
        torch.testing.assert_close(got, ref, rtol=5e-2, atol=5e-3, msg=msg)
    return "per-level norm"


def _check_frontend(name, device, strict):
    from mtvaf_amd.features import RegionFeatureCache
    im, sd, x, aux, fx = _frontend_case(name, device)
    how = set()
    for mode in ("train", "eval"):
        im.train(mode == "train")
        im.resnet.load_state_dict(sd)  # (train mode moves the running statistics)
        with torch.no_grad():
            pyr, aux_pyr = im(x, aux)
        got = torch.cat(pyr, 1).cpu()
        got_aux = torch.stack([torch.cat(a, 1) for a in aux_pyr], 1).cpu()
        how.add(_close_per_level(got, torch.from_numpy(fx[f"{mode}_main"]), name, f"{mode} main", device, strict))
        how.add(_close_per_level(got_aux, torch.from_numpy(fx[f"{mode}_aux"]), name, f"{mode} aux", device, strict))
        if mode == "train":  # BatchNorm ran in train mode, as the reference's merely requires_grad=False "frozen" trunk does
            rm, ref_rm = im.resnet.bn1.running_mean.cpu(), torch.from_numpy(fx["train_bn1_running_mean_after"])
            assert float((rm.double() - ref_rm.double()).norm() / ref_rm.double().norm()) < (1e-5 if device == "cpu" else LEVEL_TOL)
    # train-mode and eval-mode pyramids really differ (the documented deviation of the feature cache is not vacuous)
    assert float(np_rel(fx["train_main"], fx["eval_main"])) > 1e-2
    # the cache = the reference's EVAL-mode pyramid, whatever mode the model is in
    im.resnet.load_state_dict(sd)
    im.train()
    feats, fa = RegionFeatureCache(im).extract(x, aux)
    how.add(_close_per_level(feats.cpu(), torch.from_numpy(fx["eval_main"]), name, "cache main", device, strict))
    how.add(_close_per_level(fa.cpu(), torch.from_numpy(fx["eval_aux"]), name, "cache aux", device, strict))
    assert im.training
    return how


def np_rel(a, b):
    import numpy as np
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("name", ["resnet18", "resnet50"])
def test_frontend_matches_reference_golden_cpu(name):
    """The drop-in ImageModel against the pyramid the REFERENCE's ImageModel.forward produced (tests/golden/
    gen_frontend_fixture.py: reference class imported in the authoring container, seeded trunk and images): pyramid taps,
    AvgPool2d(ft // 2) pooling, aux-image order, train-mode and eval-mode BatchNorm; and RegionFeatureCache against the
    reference's eval-mode pyramid.  Plain torch on the host: no GPU needed."""
    how = _check_frontend(name, "cpu", strict=(1e-4, 1e-5))
    print(f"[frontend {name} cpu] compared: {sorted(how)}")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["resnet18", "resnet50"])
def test_frontend_matches_reference_golden_gpu(name):
    """The same on the MI355X (MIOpen picks its own convolution algorithms): every pyramid level to 2e-3 in norm, train-mode and
    eval-mode BatchNorm, main and aux images, and the feature cache (see _close_per_level)."""
    _check_frontend(name, "cuda", strict=None)


@pytest.mark.gpu
def test_cached_features_reproduce_raw_image_prefix():
    from test_model_gpu import DEV, LABELS, hf_config, make_args
    import params as P
    from mtvaf_amd.features import RegionFeatureCache
    from mtvaf_amd.models.bert_model import TVNetSAModel2
    cfg = P.EncCfg(vocab_size=64, hidden=768, heads=12, inter=128, layers=12, max_pos=32)
    args = make_args(resnet_root="random")
    args.bert_config = hf_config(cfg)
    torch.manual_seed(0)
    m = TVNetSAModel2(LABELS, None, args).to(DEV).eval()
    x = torch.randn(2, 3, 64, 64, device=DEV)
    aux = torch.randn(2, 3, 3, 64, 64, device=DEV)
    raw, _, _ = m.get_visual_prompt(x, aux, None)
    feats, fa = RegionFeatureCache(m.image_model).extract(x, aux)
    assert tuple(feats.shape) == (2, 3840, 2, 2)
    cached, _, _ = m.get_visual_prompt(feats, fa, None)
    for (k1, v1), (k2, v2) in zip(raw, cached):  # MIOpen may pick different conv algorithms between the two passes
        assert torch.allclose(k1, k2, rtol=1e-4, atol=1e-5) and torch.allclose(v1, v2, rtol=1e-4, atol=1e-5)
    assert tuple(raw[0][0].shape) == (2, 12, 16, 64)
