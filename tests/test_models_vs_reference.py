"""The nets against the reference's own (models/DispResNet.py:49-121, PoseResNet.py:14-66, resnet_encoder.py): the same
ordered state-dict keys and shapes -- "same checkpoint format" (utils.py:57-66) demonstrated, not asserted -- and, with the
weights copied across, the same outputs in train and eval mode.

torchvision is not installed in this image, and the reference's encoder subclasses ``torchvision.models.ResNet``; the
decoders are pure torch.  A TEST-ONLY stand-in for ``torchvision.models`` (the trunk / blocks of this repo's plain-torch
encoder under torchvision's names) lets the reference's modules import; what is compared is therefore the reference's
decoders, its multi-image conv1 surgery and its forward plumbing against this repo's, on the same trunk.  The key / shape
list of the reference's nets is committed as tests/golden/model_keys.json, which the repo's nets are also checked against
where /root/reference does not exist (the GPU box)."""
import importlib.util
import json
import os
import sys
import types

import pytest
import torch

import models as mine
from models import resnet_encoder as enc

REF_MODELS = "/root/reference/models"
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_keys.json")
NETS = [("DispResNet", 18), ("DispResNet", 50), ("PoseResNet", 18), ("PoseResNet", 50)]


def _reference_models():
    """Import /root/reference/models as `ref_models` behind a stand-in for torchvision.models."""
    if "ref_models" in sys.modules:
        return sys.modules["ref_models"]

    class ResNet(enc.ResNet):  # torchvision's signature: (block, layers, num_classes=1000)
        def __init__(self, block, layers, num_classes=1000):
            super().__init__(block, layers, num_classes=num_classes, num_input_images=1)

    tv, tvm, tvr = types.ModuleType("torchvision"), types.ModuleType("torchvision.models"), types.ModuleType("torchvision.models.resnet")
    tvr.BasicBlock, tvr.Bottleneck, tvr.model_urls = enc.BasicBlock, enc.Bottleneck, {}
    tvm.ResNet, tvm.resnet = ResNet, tvr
    tvm.resnet18 = lambda pretrained=False: ResNet(enc.BasicBlock, [2, 2, 2, 2])
    tvm.resnet34 = lambda pretrained=False: ResNet(enc.BasicBlock, [3, 4, 6, 3])
    tvm.resnet50 = lambda pretrained=False: ResNet(enc.Bottleneck, [3, 4, 6, 3])
    tvm.resnet101 = lambda pretrained=False: ResNet(enc.Bottleneck, [3, 4, 23, 3])
    tvm.resnet152 = lambda pretrained=False: ResNet(enc.Bottleneck, [3, 8, 36, 3])
    tv.models = tvm
    saved = {k: sys.modules.get(k) for k in ("torchvision", "torchvision.models", "torchvision.models.resnet")}
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.resnet": tvr})
    try:
        spec = importlib.util.spec_from_file_location("ref_models", os.path.join(REF_MODELS, "__init__.py"),
                                                      submodule_search_locations=[REF_MODELS])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["ref_models"] = mod
        sys.dont_write_bytecode = True
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def _keys(net):
    return [[k, list(v.shape)] for k, v in net.state_dict().items()]


def test_state_dict_layout_matches_the_committed_reference_layout():
    want = json.load(open(FIXTURE))
    for name, layers in NETS:
        net = getattr(mine, name)(layers, False)
        assert _keys(net) == want[f"{name}{layers}"], (name, layers)
        assert sum(p.numel() for p in net.parameters()) == want[f"{name}{layers}/parameters"]


needs_ref = pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="/root/reference is not mounted")


@needs_ref
def test_the_committed_layout_is_the_reference_nets():
    ref = _reference_models()
    want = json.load(open(FIXTURE))
    for name, layers in NETS:
        net = getattr(ref, name)(layers, False)
        assert _keys(net) == want[f"{name}{layers}"], (name, layers)
        assert sum(p.numel() for p in net.parameters()) == want[f"{name}{layers}/parameters"]


@needs_ref
@pytest.mark.parametrize("name,layers", NETS)
def test_same_outputs_as_the_reference_nets(name, layers):
    """Weights copied across with load_state_dict(strict=True) -- the checkpoint path of utils.py:57-66 / train.py:133-141 --
    then identical outputs in fp64: train mode (batch statistics, four disparity scales) and eval mode."""
    ref = _reference_models()
    torch.manual_seed(layers)
    a = getattr(ref, name)(layers, False).double()
    b = getattr(mine, name)(layers, False).double()
    with torch.no_grad():  # something other than the constant initialisation of the batch norms
        for p in a.parameters():
            p.add_(0.05 * torch.randn_like(p))
        for m in a.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    b.load_state_dict(a.state_dict(), strict=True)
    x = torch.randn(2, 3, 64, 96, dtype=torch.float64)
    y = torch.randn(2, 3, 64, 96, dtype=torch.float64)
    args = (x,) if name == "DispResNet" else (x, y)
    for mode in ("train", "eval"):
        getattr(a, mode)()
        getattr(b, mode)()
        oa, ob = a(*args), b(*args)
        oa, ob = (oa if isinstance(oa, (list, tuple)) else [oa]), (ob if isinstance(ob, (list, tuple)) else [ob])
        assert len(oa) == len(ob) == (4 if (name == "DispResNet" and mode == "train") else 1)
        for u, v in zip(oa, ob):
            assert u.shape == v.shape and float((u - v).abs().max()) <= 1e-12 * max(1.0, float(u.abs().max())), (mode, u.shape)
    # the running statistics moved identically in train mode
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and float((va.double() - vb.double()).abs().max()) <= 1e-12, ka


if __name__ == "__main__":  # regenerate the fixture from the reference (build container)
    ref = _reference_models()
    out = {}
    for name, layers in NETS:
        net = getattr(ref, name)(layers, False)
        out[f"{name}{layers}"] = _keys(net)
        out[f"{name}{layers}/parameters"] = sum(p.numel() for p in net.parameters())
    json.dump(out, open(FIXTURE, "w"), indent=0)
    print(FIXTURE, {k: v for k, v in out.items() if k.endswith("parameters")})
