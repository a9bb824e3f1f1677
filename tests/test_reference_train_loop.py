"""The reference's OWN training loop driven through the drop-in ("put this package in
front of the reference on PYTHONPATH and run its train.py").

/root/reference/train.py is imported unmodified with ``sc-sfmlearner-release_amd/`` first on ``sys.path``, so that its
``from loss_functions import ...`` / ``import models`` / ``import custom_transforms`` / ``from logger import ...`` /
``from datasets.sequence_folders import SequenceFolder`` (train.py:13-20) resolve to this repo; the two modules it imports
that are neither the reference's nor installed here -- ``path`` (train.py:5) and ``tensorboardX`` (:20) -- get TEST-ONLY
stand-ins, as torchvision does in tests/test_models_vs_reference.py.  Importing it also switches anomaly mode on globally
(train.py:67), which is the point: the loop below runs under it.  Then two iterations of ITS ``train()``
(train.py:235-299) and one pass of ITS ``validate_without_gt()`` (:302-362) run on a synthetic SequenceFolder tree, the
HIP kernels served by the host simulation (no GPU in the build container), and every logged loss must equal what this
repo's own ``train.train_step`` computes on the same batches with the same initial weights.

Skipped where /root/reference does not exist (the GPU box)."""
import csv
import importlib.util
import os
import pathlib
import sys
import types

import pytest
import torch

REF = "/root/reference"
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sc-sfmlearner-release_amd")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "train.py")), reason="the reference tree is not mounted")


class _Path(type(pathlib.Path())):
    """What train.py uses of path.py's Path: the / operator, open() on the result, makedirs_p()."""

    def makedirs_p(self):
        os.makedirs(str(self), exist_ok=True)
        return self


class _SummaryWriter:
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, name, value, step):
        self.scalars.append((name, float(value), int(step)))

    def add_image(self, *a, **k):
        pass


@pytest.fixture
def ref_train(monkeypatch):
    """The reference's train.py as a module, importing THIS repo's drop-in modules; anomaly mode restored afterwards."""
    assert sys.path.index(PKG) < len(sys.path)  # tests/conftest.py put the package on the path
    for name in ("loss_functions", "inverse_warp", "models", "custom_transforms", "logger", "utils", "datasets"):
        mod = sys.modules.get(name)
        assert mod is None or os.path.abspath(getattr(mod, "__file__", PKG)).startswith(PKG), (name, mod)
    stubs = {"path": types.ModuleType("path"), "tensorboardX": types.ModuleType("tensorboardX")}
    stubs["path"].Path = _Path
    stubs["tensorboardX"].SummaryWriter = _SummaryWriter
    for k, v in stubs.items():
        monkeypatch.setitem(sys.modules, k, v)
    was = torch.is_anomaly_enabled()
    spec = importlib.util.spec_from_file_location("ref_train_module", os.path.join(REF, "train.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    try:
        spec.loader.exec_module(mod)  # (runs torch.autograd.set_detect_anomaly(True), train.py:67)
        assert torch.is_anomaly_enabled()
        # its names resolved to the drop-in, not to the files next to it
        import loss_functions as LF
        assert mod.compute_photo_and_geometry_loss is LF.compute_photo_and_geometry_loss
        assert os.path.abspath(sys.modules["models"].__file__).startswith(PKG)
        yield mod
    finally:
        torch.autograd.set_detect_anomaly(was)


def test_the_references_train_and_validate_run_on_the_drop_in(ref_train, tmp_path, monkeypatch):
    from hostsim import harness
    from scsfm_hip import _lib, ops
    import custom_transforms
    import models
    import train as T
    from datasets.sequence_folders import SequenceFolder
    from datasets.synthetic import write_sequence_tree
    from logger import TermLogger
    lib = harness.lib()
    monkeypatch.setattr(_lib, "get", lambda: lib)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    monkeypatch.setattr(ref_train, "device", torch.device("cpu"))
    monkeypatch.setattr(ref_train, "n_iter", 0)
    monkeypatch.chdir(tmp_path)
    # 128 x 160 x batch 2 = 40960 mask pixels per pair-direction: ABOVE the 10000-pixel gate, all three terms are live
    H, W, B = 128, 160, 2
    root = write_sequence_tree(str(tmp_path / "data"), n_scenes=2, frames_per_scene=6, height=H, width=W)
    args = ref_train.parser.parse_args([root, "--with-pretrain", "0", "-b", str(B), "--epoch-size", "2", "--name", "t",
                                        "--print-freq", "1", "-j", "0", "--with-auto-mask", "1"])
    args.save_path = _Path(tmp_path / "ckpt").makedirs_p()
    normalize = custom_transforms.Normalize(mean=[0.45] * 3, std=[0.225] * 3)
    valid_tf = custom_transforms.Compose([custom_transforms.ArrayToTensor(), normalize])
    # (no random augmentation: both loops below must see the same batches)
    train_set = SequenceFolder(root, transform=valid_tf, seed=args.seed, train=True, sequence_length=args.sequence_length)
    val_set = SequenceFolder(root, transform=valid_tf, seed=args.seed, train=False, sequence_length=args.sequence_length)
    mk = lambda ds: torch.utils.data.DataLoader(ds, batch_size=B, shuffle=False, num_workers=0, drop_last=True)
    train_loader, val_loader = mk(train_set), mk(val_set)
    assert len(train_loader) >= 2 and len(val_loader) >= 1

    def nets():
        torch.manual_seed(3)
        d, p = models.DispResNet(18, False), models.PoseResNet(18, False)
        opt = torch.optim.Adam([{"params": d.parameters(), "lr": args.lr}, {"params": p.parameters(), "lr": args.lr}],
                               betas=(args.momentum, args.beta), weight_decay=args.weight_decay)  # train.py:188-194
        return d, p, opt

    # ---- the reference's loop ----------------------------------------------------------------------------------
    disp, pose, opt = nets()
    logger = TermLogger(n_epochs=1, train_size=2, valid_size=len(val_loader))
    writer = _SummaryWriter()
    avg = ref_train.train(args, train_loader, disp, pose, opt, args.epoch_size, logger, writer)
    rows = [[float(x) for x in r] for r in csv.reader(open(args.save_path / args.log_full), delimiter="\t")]
    assert len(rows) == 2 and all(len(r) == 4 for r in rows)            # train.py:288-290: loss, photo, smooth, geometry
    assert all(v == v and v > 0 for r in rows for v in r), rows          # every term is live and finite
    assert abs(avg - sum(r[0] for r in rows) / 2) < 1e-6
    assert [n for n, _, _ in writer.scalars][:4] == ["photometric_error", "disparity_smoothness_loss",
                                                     "geometry_consistency_loss", "total_loss"]  # (logged at i = 1)
    errors, names = ref_train.validate_without_gt(args, val_loader, disp, pose, 0, logger)
    assert names == ["Total loss", "Photo loss", "Smooth loss", "Consistency loss"]
    assert all(e == e for e in errors) and errors[1] > 0 and errors[0] == errors[1]  # train.py:352: loss = loss_1

    # ---- this repo's step on the same batches, same initial weights ----------------------------------------------
    disp2, pose2, opt2 = nets()
    disp2.train(); pose2.train()
    args2 = T.parser.parse_args([root, "--with-pretrain", "0", "-b", str(B), "--name", "t", "--with-auto-mask", "1",
                                 "--single-loss-node", "0"])
    args2.world = 1
    for i, (tgt, refs, K, _Kinv) in enumerate(train_loader):
        if i >= 2:
            break
        loss, l1, l2, l3 = T.train_step(args2, disp2, pose2, opt2, tgt, list(refs), K)[:4]
        mine = [float(t.detach()) for t in (loss, l1, l2, l3)]
        for a, b in zip(mine, rows[i]):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (i, mine, rows[i])
    for a, b in zip(list(disp.parameters()) + list(pose.parameters()), list(disp2.parameters()) + list(pose2.parameters())):
        assert torch.equal(a, b)   # the same two Adam steps, bit for bit
