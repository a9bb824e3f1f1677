"""CPU checks of the trainer's plumbing: CLI surface, model state-dict layout,
checkpoint format, dataset tree / transforms.  No GPU, no loss kernels."""
import os
import re
import sys

import numpy as np
import pytest
import torch

REF_TRAIN = "/root/reference/train.py"

REFERENCE_FLAGS = {
    "data", "--folder-type", "--sequence-length", "-j", "--workers", "--epochs", "--epoch-size", "-b", "--batch-size",
    "--lr", "--learning-rate", "--momentum", "--beta", "--weight-decay", "--wd", "--print-freq", "--seed",
    "--log-summary", "--log-full", "--log-output", "--resnet-layers", "--num-scales", "--number-of-scales", "-p",
    "--photo-loss-weight", "-s", "--smooth-loss-weight", "-c", "--geometry-consistency-weight", "--with-ssim",
    "--with-mask", "--with-auto-mask", "--with-pretrain", "--dataset", "--pretrained-disp", "--pretrained-pose",
    "--name", "--padding-mode", "--with-gt",
}


def _flags_of(path):
    src = open(path).read()
    out = set()
    for call in re.findall(r"parser\.add_argument\(([^)]*)", src):
        out.update(re.findall(r"'(-{0,2}[\w-]+)'", call.split("help=")[0].split("type=")[0].split("default=")[0]))
    return {f for f in out if f.startswith("-") or f == "data"}


def test_cli_has_every_reference_flag():
    import train
    mine = set()
    for a in train.parser._actions:
        mine.update(a.option_strings or [a.dest])
    assert REFERENCE_FLAGS <= mine, REFERENCE_FLAGS - mine
    if os.path.exists(REF_TRAIN):  # the list above is the reference's (train.py:27-61)
        assert _flags_of(REF_TRAIN) - {"choices"} <= mine | {"data"}
    d = train.parser.parse_args(["/tmp/x", "--name", "n"])
    assert (d.batch_size, d.lr, d.momentum, d.beta, d.sequence_length, d.num_scales) == (4, 1e-4, 0.9, 0.999, 3, 1)
    assert (d.photo_loss_weight, d.smooth_loss_weight, d.geometry_consistency_weight) == (1, 0.1, 0.5)
    assert (d.with_ssim, d.with_mask, d.with_auto_mask, d.padding_mode, d.dataset) == (1, 1, 0, "zeros", "kitti")


def test_models_have_the_reference_state_dict_layout():
    import models
    disp, pose = models.DispResNet(18, False), models.PoseResNet(18, False)
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(disp) == 14842236 and count(pose) == 13011950  # 14.84 M / 13.01 M
    kd, kp = list(disp.state_dict()), list(pose.state_dict())
    assert "encoder.encoder.conv1.weight" in kd and "encoder.encoder.fc.weight" in kd
    assert "encoder.encoder.layer4.1.bn2.running_var" in kd and "encoder.encoder.layer2.0.downsample.0.weight" in kd
    assert [k for k in kd if k.startswith("decoder")][:2] == ["decoder.decoder.0.conv.conv.weight", "decoder.decoder.0.conv.conv.bias"]
    assert kd[-2:] == ["decoder.decoder.13.conv.weight", "decoder.decoder.13.conv.bias"]
    assert [k for k in kp if k.startswith("decoder")] == [f"decoder.net.{i}.{w}" for i in range(4) for w in ("weight", "bias")]
    assert tuple(pose.state_dict()["encoder.encoder.conv1.weight"].shape) == (64, 6, 7, 7)
    d50 = models.DispResNet(50, False)
    assert "encoder.encoder.layer1.0.conv3.weight" in d50.state_dict()
    assert tuple(d50.state_dict()["decoder.decoder.0.conv.conv.weight"].shape) == (256, 2048, 3, 3)
    # train mode: four scales, disparity in (0.01, 10.01); eval mode: scale 0 only (DispResNet.py:118-121)
    disp.train()
    outs = disp(torch.randn(1, 3, 64, 96))
    assert [tuple(o.shape[-2:]) for o in outs] == [(64, 96), (32, 48), (16, 24), (8, 12)]
    assert float(outs[0].min()) > 0.01 and float(outs[0].max()) < 10.01
    disp.eval()
    assert tuple(disp(torch.randn(1, 3, 64, 96)).shape) == (1, 1, 64, 96)
    p = pose(torch.randn(2, 3, 64, 96), torch.randn(2, 3, 64, 96))
    assert tuple(p.shape) == (2, 6) and float(p.abs().max()) < 1.0
    # the unused classifier head stays in the state dict but never asks for a gradient (DDP)
    assert not disp.encoder.encoder.fc.weight.requires_grad
    with pytest.raises(RuntimeError, match="offline"):
        models.DispResNet(18, True)


def test_checkpoint_format_round_trip(tmp_path):
    import models
    from utils import save_checkpoint
    disp, pose = models.DispResNet(18, False), models.PoseResNet(18, False)
    save_checkpoint(tmp_path, {"epoch": 3, "state_dict": disp.state_dict()}, {"epoch": 3, "state_dict": pose.state_dict()}, True)
    names = sorted(os.listdir(tmp_path))
    assert names == ["dispnet_checkpoint.pth.tar", "dispnet_model_best.pth.tar", "exp_pose_checkpoint.pth.tar",
                     "exp_pose_model_best.pth.tar"]
    blob = torch.load(tmp_path / "dispnet_model_best.pth.tar")
    assert set(blob) == {"epoch", "state_dict"} and blob["epoch"] == 3
    models.DispResNet(18, False).load_state_dict(blob["state_dict"], strict=True)  # test_disp.py:41 loads strictly
    models.PoseResNet(18, False).load_state_dict(torch.load(tmp_path / "exp_pose_checkpoint.pth.tar")["state_dict"], strict=True)


def test_dataset_tree_and_transforms(tmp_path):
    import custom_transforms as T
    from datasets.sequence_folders import SequenceFolder
    from datasets.synthetic import InMemorySequences, write_sequence_tree
    from datasets.validation_folders import ValidationSet
    root = write_sequence_tree(str(tmp_path / "kitti"), n_scenes=2, frames_per_scene=5, height=64, width=96)
    tf = T.Compose([T.RandomHorizontalFlip(), T.RandomScaleCrop(), T.ArrayToTensor(), T.Normalize([0.45] * 3, [0.225] * 3)])
    ds = SequenceFolder(root, transform=tf, seed=0, train=True, sequence_length=3)
    assert len(ds) == 3  # 5 frames -> 3 centred triplets, one train scene
    tgt, refs, K, Kinv = ds[0]
    assert tuple(tgt.shape) == (3, 64, 96) and len(refs) == 2 and K.shape == (3, 3)
    np.testing.assert_allclose(K @ Kinv, np.eye(3), atol=1e-4)
    assert -2.1 < float(tgt.min()) and float(tgt.max()) < 2.5  # (x/255 - .45)/.225
    loader = torch.utils.data.DataLoader(ds, batch_size=2)
    b_tgt, b_refs, b_K, _ = next(iter(loader))
    assert tuple(b_tgt.shape) == (2, 3, 64, 96) and len(b_refs) == 2 and tuple(b_K.shape) == (2, 3, 3)
    val = ValidationSet(root, transform=T.Compose([T.ArrayToTensor(), T.Normalize([0.45] * 3, [0.225] * 3)]), dataset="kitti")
    img, depth = val[0]
    assert tuple(img.shape) == (3, 64, 96) and tuple(depth.shape) == (64, 96)
    mem = InMemorySequences(4, 32, 48, sequence_length=5)
    t, r, K, _ = mem[1]
    assert tuple(t.shape) == (3, 32, 48) and len(r) == 4
    # flip moves the principal point (custom_transforms.py:54-56)
    import random
    random.seed(1)
    K0 = np.array([[100., 0, 30], [0, 100, 20], [0, 0, 1]], dtype=np.float32)
    while True:
        out, K1 = T.RandomHorizontalFlip()([np.zeros((40, 64, 3), np.float32)], K0)
        if K1[0, 2] != K0[0, 2]:
            assert K1[0, 2] == 64 - 30
            break


def test_pair_folder_follows_the_reference_layout(tmp_path):
    """datasets/pair_folders.py:33-45 of the reference: sorted frames, pair k = frames (2k, 2k+1) in that order with the
    k-th intrinsics file of the scene; never swapped."""
    from PIL import Image
    from datasets.pair_folders import PairFolder
    scene = tmp_path / "s0"
    scene.mkdir()
    for k in range(3):
        for j in range(2):
            Image.fromarray(np.full((8, 12, 3), 40 * k + 10 * j, dtype=np.uint8)).save(scene / f"{k:04d}_{j}.jpg", quality=100)
        np.savetxt(scene / f"{k:04d}_cam.txt", np.array([[100.0 + k, 0, 6], [0, 100.0 + k, 4], [0, 0, 1]]))
    Image.fromarray(np.zeros((8, 12, 3), dtype=np.uint8)).save(scene / "9999_0.jpg")  # unpaired trailing frame: ignored
    (tmp_path / "train.txt").write_text("s0\n")
    ds = PairFolder(str(tmp_path), seed=0, train=True, transform=None)
    assert len(ds) == 3
    seen = set()
    for i in range(3):
        tgt, refs, K, Kinv = ds[i]
        k = int(round(float(K[0, 0]) - 100.0))
        seen.add(k)
        assert len(refs) == 1
        assert abs(float(tgt.mean()) - 40 * k) < 3 and abs(float(refs[0].mean()) - (40 * k + 10)) < 3  # (0 then 1, fixed)
        np.testing.assert_allclose(K @ Kinv, np.eye(3), atol=1e-5)
    assert seen == {0, 1, 2}


def test_train_and_validation_loops_run_on_the_host_simulation(tmp_path, monkeypatch):
    """train.train + train.validate_without_gt -- data loader, augmentation chain, nets, the loss path (HIP kernels via
    tests/hostsim), optimiser, progress logging, checkpoint -- for two iterations on the synthetic data set.  The test
    swaps the library for the simulator from OUTSIDE; the product has no such switch."""
    import argparse
    from hostsim import harness
    from scsfm_hip import _lib, config as hip_config, ops
    import train as T
    from logger import TermLogger
    from utils import save_checkpoint
    lib = harness.lib()
    monkeypatch.setattr(_lib, "get", lambda: lib)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    monkeypatch.setattr(T, "device", torch.device("cpu"))
    monkeypatch.chdir(tmp_path)
    args = T.parser.parse_args(["synthetic:8:64x96", "--with-pretrain", "0", "-b", "2", "--epoch-size", "2", "--epochs", "1",
                                "-j", "0", "--name", "t", "--print-freq", "1"])
    args.save_path = str(tmp_path)
    args.world = 1
    for name in (args.log_summary, args.log_full):
        open(os.path.join(args.save_path, name), "w").close()
    torch.manual_seed(0)
    train_set, val_set = T.build_datasets(args)
    mk = lambda ds, shuffle: torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=shuffle, num_workers=0)
    train_loader, val_loader = mk(train_set, True), mk(val_set, False)
    import models
    disp, pose = models.DispResNet(18, False), models.PoseResNet(18, False)
    opt = torch.optim.Adam([{"params": disp.parameters()}, {"params": pose.parameters()}], lr=1e-4)
    hip_config.set_weight_hint(args.photo_loss_weight, args.geometry_consistency_weight)
    logger = TermLogger(n_epochs=1, train_size=2, valid_size=len(val_loader))
    before = [p.detach().clone() for p in list(disp.parameters())[:3]]
    loss = T.train(args, train_loader, disp, pose, opt, args.epoch_size, logger, T._ScalarLog())
    assert loss == loss  # finite (64 x 96 x 2 pixels are below the 10000-pixel gate: the smooth term alone drives the step)
    assert any(not torch.equal(a, b) for a, b in zip(before, list(disp.parameters())[:3]))
    # one more epoch with all four scales of the decoder: the coarser maps reach the kernels as they are (depth_shift)
    args.num_scales = 4
    logger.reset_train_bar()
    loss4 = T.train(args, train_loader, disp, pose, opt, 1, logger, T._ScalarLog())
    assert loss4 == loss4
    args.num_scales = 1
    errors, names = T.validate_without_gt(args, val_loader, disp, pose, 0, logger)
    assert len(errors) == 4 and names[0] == "Total loss" and all(e == e for e in errors)
    rows = open(os.path.join(args.save_path, args.log_full)).read().strip().split("\n")
    assert len(rows) == 3 and len(rows[0].split("\t")) == 4
    save_checkpoint(args.save_path, {"epoch": 1, "state_dict": disp.state_dict()}, {"epoch": 1, "state_dict": pose.state_dict()}, True)
    assert os.path.exists(os.path.join(args.save_path, "dispnet_model_best.pth.tar"))
