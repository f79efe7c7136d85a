"""Front-end fixture: the REFERENCE's own `ImageModel.forward` / `get_resnet_prompt`
(models/bert_model.py:63-111, imported from /root/reference in the authoring container through the shim of
gen_golden.py) run on seeded ResNet trunk weights and seeded images, in train mode (what the reference does while
training: `requires_grad=False` only, BatchNorm keeps batch statistics -- modules/train.py:579, 920-921) and in eval
mode (dev / test time; what `mtvaf_amd.features.RegionFeatureCache` caches).

torchvision is absent from the image, so the reference's `torchvision.models.resnetNN` constructors are supplied by
`mtvaf_amd/models/resnet.py` (torchvision's published architecture and state_dict names; tests/test_frontend.py checks
parameter counts and key names).  What this fixture pins is therefore everything the reference does AROUND the trunk --
which children run, where the pyramid taps sit, the `AvgPool2d(ft // 2)` pooling, the aux-image permutation, the BatchNorm
mode -- not the trunk arithmetic against torchvision itself (parity unpinned there: package and weights absent).

    python tests/golden/gen_frontend_fixture.py       # writes tests/golden/frontend_resnet{18,50}.npz

Weights and images are regenerated from numpy PCG64 seeds (params.resnet_params / params.image_batch): the fixture
holds the expected pyramids only.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import gen_golden as G  # noqa: E402
import params as P  # noqa: E402

CASES = {"resnet18": dict(seed=301, B=2, n_aux=3, hw=64, kw=dict(use_18=True)),
         "resnet50": dict(seed=311, B=2, n_aux=3, hw=64, kw=dict())}


def main():
    G.install_shim()
    import torchvision.models as tvm
    from mtvaf_amd.models import resnet as our_resnet
    for n in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(tvm, n, (lambda ctor: (lambda pretrained=False, **k: ctor()))(getattr(our_resnet, n)))
    import models.bert_model as rbm  # the reference
    tmp = tempfile.mkdtemp(prefix="mtvaf_frontend_fixture_")
    for name, c in CASES.items():
        sd = P.resnet_params(getattr(our_resnet, name)(), c["seed"])
        torch.save(sd, os.path.join(tmp, f"{name}.pth"))
        ref = rbm.ImageModel(resnet_root=tmp, **c["kw"])
        x, aux = P.image_batch(c["seed"] + 1, c["B"], c["n_aux"], c["hw"])
        out = {"seed": c["seed"], "B": c["B"], "n_aux": c["n_aux"], "hw": c["hw"]}
        for mode in ("train", "eval"):
            ref.train(mode == "train")
            ref.load_state_dict({"resnet." + k: v for k, v in sd.items()})  # (train mode moves the running statistics)
            with torch.no_grad():
                pyr, aux_pyr = ref(x, aux)
            assert len(pyr) == 4 and len(aux_pyr) == c["n_aux"]
            out[f"{mode}_main"] = torch.cat(pyr, 1).numpy()                        # [B, F, 2, 2]
            out[f"{mode}_aux"] = torch.stack([torch.cat(a, 1) for a in aux_pyr], 1).numpy()  # [B, n_aux, F, 2, 2]
            if mode == "train":  # the running statistics after the (1 + n_aux) train-mode passes: BatchNorm really ran in train mode
                out["train_bn1_running_mean_after"] = ref.resnet.bn1.running_mean.numpy().copy()
        path = os.path.join(HERE, f"frontend_{name}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, {k: getattr(v, "shape", v) for k, v in out.items()}, os.path.getsize(path))


if __name__ == "__main__":
    main()
