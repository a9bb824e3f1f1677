"""SC-SfMLearner training, MI355X-native: the reference's command line (train.py:24-61) and checkpoint
format, the HIP loss path (loss_functions.py / inverse_warp.py of this directory), DispResNet /
PoseResNet on PyTorch-ROCm (MIOpen), and pure data parallelism with one process per GPU:

    python train.py DATA --resnet-layers 18 --num-scales 1 -b12 -s0.1 -c0.5 --sequence-length 3 \
        --with-ssim 1 --with-mask 1 --with-auto-mask 1 --with-pretrain 0 --name r18            # 1 GPU
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py DATA ... -b12       # 8 GPUs

Differences from the reference, all forced by the platform or by the scale-out design:
  * `nn.DataParallel` (train.py:168-169: one process, loss on GPU 0 over the gathered batch) becomes
    `DistributedDataParallel` over RCCL; `-b` is the PER-GPU batch (global = world x b); the loss is
    computed on each rank's shard.  `--exact-mask-normalisation` (extension) all-reduces the mask sums
    so that the loss equals the single-process loss on the global batch.
  * `--with-pretrain 1` needs ImageNet weights from the network -> rejected offline; use
    `--pretrained-disp/--pretrained-pose`.
  * DATA may be `synthetic:N:HxW` (extension): N in-memory random sequences, for smoke / throughput runs.
  * tensorboardX / blessings / progressbar2 / path are not installed: scalars go to the two TSV logs
    (same files and columns as the reference) and to stdout.
  * `torch.autograd.set_detect_anomaly(True)` (train.py:67) is not enabled: the kernels are NaN-free.
"""
import argparse
import csv
import datetime
import os
import time

import numpy as np
import torch
import torch.optim
import torch.utils.data

import custom_transforms
import models
from logger import AverageMeter, TermLogger
from loss_functions import compute_errors, compute_photo_and_geometry_loss, compute_smooth_loss
try:  # an extension of this repo's loss_functions; absent when the reference's module is on the path
    from loss_functions import compute_total_loss
except ImportError:
    compute_total_loss = None
from scsfm_hip import config as hip_config
from scsfm_hip import dist as hip_dist
from utils import save_checkpoint

parser = argparse.ArgumentParser(description='Structure from Motion Learner training on KITTI and CityScapes Dataset',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('data', metavar='DIR', help='path to dataset')
parser.add_argument('--folder-type', type=str, choices=['sequence', 'pair'], default='sequence', help='the dataset dype to train')
parser.add_argument('--sequence-length', type=int, metavar='N', help='sequence length for training', default=3)
parser.add_argument('-j', '--workers', default=4, type=int, metavar='N', help='number of data loading workers')
parser.add_argument('--epochs', default=200, type=int, metavar='N', help='number of total epochs to run')
parser.add_argument('--epoch-size', default=0, type=int, metavar='N', help='manual epoch size (will match dataset size if not set)')
parser.add_argument('-b', '--batch-size', default=4, type=int, metavar='N', help='mini-batch size (per GPU)')
parser.add_argument('--lr', '--learning-rate', default=1e-4, type=float, metavar='LR', help='initial learning rate')
parser.add_argument('--momentum', default=0.9, type=float, metavar='M', help='momentum for sgd, alpha parameter for adam')
parser.add_argument('--beta', default=0.999, type=float, metavar='M', help='beta parameters for adam')
parser.add_argument('--weight-decay', '--wd', default=0, type=float, metavar='W', help='weight decay')
parser.add_argument('--print-freq', default=10, type=int, metavar='N', help='print frequency')
parser.add_argument('--seed', default=0, type=int, help='seed for random functions, and network initialization')
parser.add_argument('--log-summary', default='progress_log_summary.csv', metavar='PATH', help='csv where to save per-epoch train and valid stats')
parser.add_argument('--log-full', default='progress_log_full.csv', metavar='PATH', help='csv where to save per-gradient descent train stats')
parser.add_argument('--log-output', action='store_true', help='will log dispnet outputs at validation step')
parser.add_argument('--resnet-layers', type=int, default=18, choices=[18, 50], help='number of ResNet layers for depth estimation.')
parser.add_argument('--num-scales', '--number-of-scales', type=int, help='the number of scales', metavar='W', default=1)
parser.add_argument('-p', '--photo-loss-weight', type=float, help='weight for photometric loss', metavar='W', default=1)
parser.add_argument('-s', '--smooth-loss-weight', type=float, help='weight for disparity smoothness loss', metavar='W', default=0.1)
parser.add_argument('-c', '--geometry-consistency-weight', type=float, help='weight for depth consistency loss', metavar='W', default=0.5)
parser.add_argument('--with-ssim', type=int, default=1, help='with ssim or not')
parser.add_argument('--with-mask', type=int, default=1, help='with the the mask for moving objects and occlusions or not')
parser.add_argument('--with-auto-mask', type=int, default=0, help='with the the mask for stationary points')
parser.add_argument('--with-pretrain', type=int, default=1, help='with or without imagenet pretrain for resnet')
parser.add_argument('--dataset', type=str, choices=['kitti', 'nyu'], default='kitti', help='the dataset to train')
parser.add_argument('--pretrained-disp', dest='pretrained_disp', default=None, metavar='PATH', help='path to pre-trained dispnet model')
parser.add_argument('--pretrained-pose', dest='pretrained_pose', default=None, metavar='PATH', help='path to pre-trained Pose net model')
parser.add_argument('--name', dest='name', type=str, required=True, help='name of the experiment, checkpoints are stored in checpoints/name')
parser.add_argument('--padding-mode', type=str, choices=['zeros', 'border'], default='zeros',
                    help='padding mode for image warping : this is important for photometric differenciation when going outside target image.'
                         ' zeros will null gradients outside target image.'
                         ' border will only null gradients of the coordinate outside (x or y)')
parser.add_argument('--with-gt', action='store_true', help='use ground truth for validation. \
                    You need to store it in npy 2D arrays see data/kitti_raw_loader.py for an example')
# extensions (not in the reference)
parser.add_argument('--gpu-augment', action='store_true',
                    help='run flip / zoom-crop / normalisation on the GPU (byte-exact with the PIL transform chain) '
                         'instead of in the data-loader workers')
parser.add_argument('--single-loss-node', type=int, default=1,
                    help='1: the two losses and their weighted sum behind one autograd node (extension; same values and '
                         'gradients, fewer launches), 0: the three reference-style calls')
parser.add_argument('--channels-last', type=int, default=0,
                    help='1: both nets (weights and activations) in NHWC memory format -- MIOpen\'s fp32 igemm solvers then need '
                         'no NCHW<->NHWC transposes around them; the loss path keeps reading the NCHW batch.  Off by default: '
                         'other solvers, other rounding (losses equal to ~1e-6, not bit for bit)')
parser.add_argument('--rank-affinity', default='auto', choices=['auto', 'off'],
                    help='data parallel: auto = every rank (and the loader workers it forks) is confined to its share of the '
                         'host cores and -j is capped to that share; off = leave placement to the operating system')
parser.add_argument('--exact-mask-normalisation', action='store_true',
                    help='data parallel: all-reduce the mask sums so the loss equals the single-process loss on the global batch. '
                         '(BatchNorm running statistics stay per rank either way -- as per replica under the reference\'s '
                         'nn.DataParallel, train.py:168-169 -- and rank 0\'s are the ones validated and checkpointed.)')

best_error = -1
n_iter = 0
device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class _ScalarLog(object):
    """Stand-in for tensorboardX.SummaryWriter (train.py:85-89 of the reference; the package is absent here): keeps
    the call sites and drops the data -- saying so once on stdout.  The two TSV logs (progress_log_*.csv) carry
    the same scalars."""
    _announced = False

    def _announce(self):
        if not _ScalarLog._announced:
            _ScalarLog._announced = True
            print("=> tensorboardX is not installed: TensorBoard scalars / images are not written "
                  "(losses are in progress_log_full.csv and progress_log_summary.csv)")

    def add_scalar(self, *a, **k):
        self._announce()

    def add_image(self, *a, **k):
        self._announce()


def build_datasets(args):
    normalize = custom_transforms.Normalize(mean=[0.45, 0.45, 0.45], std=[0.225, 0.225, 0.225])
    train_transform = custom_transforms.Compose([custom_transforms.RandomHorizontalFlip(), custom_transforms.RandomScaleCrop(),
                                                 custom_transforms.ArrayToTensor(), normalize])
    valid_transform = custom_transforms.Compose([custom_transforms.ArrayToTensor(), normalize])
    if args.gpu_augment:  # workers only decode; the transform chain runs in scsfm_hip.augment
        train_transform = custom_transforms.Compose([custom_transforms.ArrayToUint8()])
    if args.data.startswith('synthetic:'):
        from datasets.synthetic import InMemorySequences
        _, n, hw = args.data.split(':')
        h, w = (int(v) for v in hw.lower().split('x'))
        return (InMemorySequences(int(n), h, w, args.sequence_length, seed=args.seed),
                InMemorySequences(max(int(n) // 4, args.batch_size), h, w, args.sequence_length, seed=args.seed + 1))
    from datasets.sequence_folders import SequenceFolder
    if args.folder_type == 'sequence':
        train_set = SequenceFolder(args.data, transform=train_transform, seed=args.seed, train=True,
                                   sequence_length=args.sequence_length, dataset=args.dataset)
    else:
        from datasets.pair_folders import PairFolder
        train_set = PairFolder(args.data, seed=args.seed, train=True, transform=train_transform)
    if args.with_gt:
        from datasets.validation_folders import ValidationSet
        val_set = ValidationSet(args.data, transform=valid_transform, dataset=args.dataset)
    else:
        val_set = SequenceFolder(args.data, transform=valid_transform, seed=args.seed, train=False,
                                 sequence_length=args.sequence_length, dataset=args.dataset)
    return train_set, val_set


def main():
    global best_error, n_iter, device
    args = parser.parse_args()
    if args.with_pretrain:
        # the reference's default (--with-pretrain 1, train.py:53) downloads ImageNet weights; say so before any
        # dataset is crawled rather than from inside model construction
        from models.resnet_encoder import imagenet_weights_path
        missing = [n for n in sorted({args.resnet_layers, 18}) if imagenet_weights_path(n) is None]
        if missing:
            parser.error("--with-pretrain 1 needs ImageNet weights, which cannot be downloaded here: put "
                         + ", ".join("resnet{}.pth".format(n) for n in missing)
                         + " (torchvision state dicts) in a directory named by SCSFM_IMAGENET_WEIGHTS, or pass "
                           "--with-pretrain 0 (optionally with --pretrained-disp / --pretrained-pose)")
    rank, local_rank, world = hip_dist.init_process_group_from_env()
    is_main = rank == 0
    if world > 1 and args.rank_affinity == 'auto':
        # eight ranks on one host: each on its own block of cores, loader workers included (they inherit the mask)
        block = hip_dist.pin_rank_to_its_cores(local_rank)
        workers = hip_dist.loader_workers_for_rank(args.workers, block)
        if block is not None:
            print("=> rank {}: cores {}-{} ({}), {} intra-op threads, {} loader workers{}".format(
                rank, block[0], block[-1], len(block), torch.get_num_threads(), workers,
                " (capped from -j {})".format(args.workers) if workers != args.workers else ""))
        args.workers = workers
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        raise SystemExit("train.py needs a HIP device: the loss path has no CPU fallback")

    timestamp = datetime.datetime.now().strftime("%m-%d-%H:%M")
    args.save_path = os.path.join('checkpoints', args.name, timestamp)
    if is_main:
        print('=> will save everything to {}'.format(args.save_path))
        os.makedirs(args.save_path, exist_ok=True)

    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    # the reference turns the conv auto-tuner on (train.py:83).  On MIOpen that flag means an exhaustive per-layer search
    # -- every applicable solver compiled and timed for every convolution shape of both nets: more than 13 minutes before
    # the first step at 256x832 -- and the step it buys is not faster (tools/e2e_probe.py), so it is off unless asked for
    torch.backends.cudnn.benchmark = os.environ.get("SCSFM_CUDNN_BENCHMARK", "0") == "1"
    if is_main:
        print("=> torch.backends.cudnn.benchmark = {} (the reference sets True, train.py:83; on MIOpen that is an exhaustive "
              "per-layer solver search: SCSFM_CUDNN_BENCHMARK=1 turns it on)".format(torch.backends.cudnn.benchmark))

    training_writer = _ScalarLog()
    output_writers = [_ScalarLog() for _ in range(3)] if args.log_output else []

    if is_main:
        print("=> fetching scenes in '{}'".format(args.data))
    train_set, val_set = build_datasets(args)
    if is_main:
        print('{} samples found in {} train scenes'.format(len(train_set), len(train_set.scenes)))
        print('{} samples found in {} valid scenes'.format(len(val_set), len(val_set.scenes)))
    train_sampler = torch.utils.data.distributed.DistributedSampler(train_set, world, rank, shuffle=True, seed=args.seed,
                                                                    drop_last=True) if world > 1 else None
    train_loader = torch.utils.data.DataLoader(train_set, batch_size=args.batch_size, shuffle=train_sampler is None,
                                               sampler=train_sampler, num_workers=args.workers, pin_memory=True,
                                               drop_last=world > 1)
    val_loader = torch.utils.data.DataLoader(val_set, batch_size=args.batch_size, shuffle=False,
                                             num_workers=args.workers, pin_memory=True)
    if args.epoch_size == 0:
        args.epoch_size = len(train_loader)

    if is_main:
        print("=> creating model")
    disp_net = models.DispResNet(args.resnet_layers, args.with_pretrain).to(device)
    pose_net = models.PoseResNet(18, args.with_pretrain).to(device)
    if args.channels_last:
        disp_net, pose_net = disp_net.to(memory_format=torch.channels_last), pose_net.to(memory_format=torch.channels_last)
    if args.pretrained_disp:
        if is_main:
            print("=> using pre-trained weights for DispResNet")
        disp_net.load_state_dict(torch.load(args.pretrained_disp, map_location=device)['state_dict'], strict=False)
    if args.pretrained_pose:
        if is_main:
            print("=> using pre-trained weights for PoseResNet")
        pose_net.load_state_dict(torch.load(args.pretrained_pose, map_location=device)['state_dict'], strict=False)

    if world > 1:
        # gradients: one bucketed all-reduce per step over RCCL / xGMI, overlapped with backward
        freeze_unused_scale_heads(disp_net, args.num_scales)
        disp_net, pose_net = wrap_ddp(disp_net, local_rank), wrap_ddp(pose_net, local_rank)
        if args.exact_mask_normalisation:
            hip_dist.enable_exact_normalisation()
    args.world = world
    # let the loss speculate on the weights its two outputs are multiplied with (scsfm_hip/config.py)
    hip_config.set_weight_hint(args.photo_loss_weight * (world if args.exact_mask_normalisation and world > 1 else 1),
                               args.geometry_consistency_weight * (world if args.exact_mask_normalisation and world > 1 else 1))

    if is_main:
        print('=> setting adam solver')
    optim_params = [{'params': [p for p in disp_net.parameters() if p.requires_grad], 'lr': args.lr},
                    {'params': [p for p in pose_net.parameters() if p.requires_grad], 'lr': args.lr}]
    optimizer = torch.optim.Adam(optim_params, betas=(args.momentum, args.beta), weight_decay=args.weight_decay)

    if is_main:
        with open(os.path.join(args.save_path, args.log_summary), 'w') as csvfile:
            csv.writer(csvfile, delimiter='\t').writerow(['train_loss', 'validation_loss'])
        with open(os.path.join(args.save_path, args.log_full), 'w') as csvfile:
            csv.writer(csvfile, delimiter='\t').writerow(['train_loss', 'photo_loss', 'smooth_loss', 'geometry_consistency_loss'])

    logger = TermLogger(n_epochs=args.epochs, train_size=min(len(train_loader), args.epoch_size), valid_size=len(val_loader))
    logger.epoch_bar.start()
    for epoch in range(args.epochs):
        logger.epoch_bar.update(epoch)
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        logger.reset_train_bar()
        train_loss = train(args, train_loader, disp_net, pose_net, optimizer, args.epoch_size, logger, training_writer, is_main)
        if is_main:
            logger.train_writer.write(' * Avg Loss : {:.3f}'.format(train_loss))

        # validation and checkpoints on rank 0 (the nets are identical on every rank after the step)
        if is_main:
            logger.reset_valid_bar()
            d_net = disp_net.module if world > 1 else disp_net
            p_net = pose_net.module if world > 1 else pose_net
            if args.with_gt:
                errors, error_names = validate_with_gt(args, val_loader, d_net, epoch, logger, output_writers)
            else:
                errors, error_names = validate_without_gt(args, val_loader, d_net, p_net, epoch, logger, output_writers)
            logger.valid_writer.write(' * Avg {}'.format(', '.join('{} : {:.3f}'.format(n, e) for n, e in zip(error_names, errors))))
            for error, name in zip(errors, error_names):
                training_writer.add_scalar(name, error, epoch)
            decisive_error = errors[1]
            if best_error < 0:
                best_error = decisive_error
            is_best = decisive_error < best_error
            best_error = min(best_error, decisive_error)
            save_checkpoint(args.save_path, {'epoch': epoch + 1, 'state_dict': d_net.state_dict()},
                            {'epoch': epoch + 1, 'state_dict': p_net.state_dict()}, is_best)
            with open(os.path.join(args.save_path, args.log_summary), 'a') as csvfile:
                csv.writer(csvfile, delimiter='\t').writerow([train_loss, decisive_error])
        if world > 1:
            torch.distributed.barrier()
    logger.epoch_bar.finish()
    if world > 1:
        torch.distributed.destroy_process_group()


def freeze_unused_scale_heads(disp_net, num_scales):
    """In train mode DispResNet returns four scales (DispResNet.py:118-119) but the loss reads only the first
    ``--num-scales`` of them (loss_functions.py:55; every reference script passes 1), so the output convolutions
    of the other scales never receive a gradient.  Single-process that is harmless (Adam skips parameters
    whose .grad is None); DistributedDataParallel, however, waits for every trainable parameter and raises at the
    second iteration.  Marking those heads as not trainable changes nothing numerically -- they stay in the
    state dict at their initial values exactly as in the reference -- and keeps DDP's bucket bookkeeping static."""
    net = disp_net.module if hasattr(disp_net, "module") else disp_net
    dec = net.decoder
    for scale, idx in dec._head.items():
        if scale >= num_scales:
            for p in dec.decoder[idx].parameters():
                p.requires_grad_(False)


def wrap_ddp(net, local_rank=None):
    """One process per GPU: bucketed all-reduce of the gradients (RCCL over xGMI; gloo on CPU), overlapped with
    backward.  ``broadcast_buffers=False``: each net runs several forwards per step before the one backward
    (3x DispResNet, 4x PoseResNet at sequence length 3, train.py:426-444) and DDP's default re-broadcast of the
    BatchNorm running statistics at the start of every forward rewrites, in place, buffers the earlier forwards
    saved for their backward.  BatchNorm statistics are per replica, as under the reference's nn.DataParallel
    (train.py:168-169); rank 0's are what is checkpointed."""
    ddp = torch.nn.parallel.DistributedDataParallel
    ids = [local_rank] if (local_rank is not None and next(net.parameters()).is_cuda) else None
    return ddp(net, device_ids=ids, bucket_cap_mb=64, gradient_as_bucket_view=True, broadcast_buffers=False)


def train_step(args, disp_net, pose_net, optimizer, tgt_img, ref_imgs, intrinsics):
    """One optimisation step on device tensors (train.py:258-282).  Returns the four losses (tensors).

    Exact mask normalisation at world > 1: every rank holds the GLOBAL photo / geometry losses (with gradients
    w.r.t. its own shard) and DDP averages gradients, so those two terms enter the differentiated sum multiplied
    by the world size; the smooth term is a per-shard mean whose average over ranks already is the global one.
    The returned ``loss`` is the unscaled w1*l1 + w2*l2 + w3*l3."""
    w1, w2, w3 = args.photo_loss_weight, args.smooth_loss_weight, args.geometry_consistency_weight
    scale = getattr(args, 'world', 1) if getattr(args, 'exact_mask_normalisation', False) else 1
    if getattr(args, 'channels_last', 0):  # the nets read NHWC copies; the loss path below keeps the NCHW batch
        net_tgt = tgt_img.contiguous(memory_format=torch.channels_last)
        net_refs = [r.contiguous(memory_format=torch.channels_last) for r in ref_imgs]
    else:
        net_tgt, net_refs = tgt_img, ref_imgs
    tgt_depth, ref_depths = compute_depth(disp_net, net_tgt, net_refs)
    poses, poses_inv = compute_pose_with_inv(pose_net, net_tgt, net_refs)
    if getattr(args, 'single_loss_node', 1) and compute_total_loss is not None:
        # the three lines of the reference below as one autograd node (same values and gradients, fewer launches)
        objective, loss_1, loss_2, loss_3 = compute_total_loss(tgt_img, ref_imgs, intrinsics, tgt_depth, ref_depths, poses,
                                                               poses_inv, args.num_scales, args.with_ssim, args.with_mask,
                                                               args.with_auto_mask, args.padding_mode, w1 * scale, w2,
                                                               w3 * scale)
    else:  # train.py:259-268 verbatim
        loss_1, loss_3 = compute_photo_and_geometry_loss(tgt_img, ref_imgs, intrinsics, tgt_depth, ref_depths, poses,
                                                         poses_inv, args.num_scales, args.with_ssim, args.with_mask,
                                                         args.with_auto_mask, args.padding_mode)
        loss_2 = compute_smooth_loss(tgt_depth, tgt_img, ref_depths, ref_imgs)
        objective = (w1 * scale) * loss_1 + w2 * loss_2 + (w3 * scale) * loss_3
    optimizer.zero_grad(set_to_none=True)
    objective.backward()
    optimizer.step()
    if scale == 1:
        loss = objective
    else:
        with torch.no_grad():
            loss = w1 * loss_1.detach() + w2 * loss_2.detach() + w3 * loss_3.detach()
    return loss, loss_1, loss_2, loss_3


def train(args, train_loader, disp_net, pose_net, optimizer, epoch_size, logger, train_writer, is_main=True):
    global n_iter, device
    batch_time = AverageMeter()
    data_time = AverageMeter()
    losses = AverageMeter(precision=4)
    disp_net.train()
    pose_net.train()
    end = time.time()
    logger.train_bar.update(0)
    for i, (tgt_img, ref_imgs, intrinsics, intrinsics_inv) in enumerate(train_loader):
        log_losses = i > 0 and n_iter % args.print_freq == 0
        data_time.update(time.time() - end)
        tgt_img = tgt_img.to(device, non_blocking=True)
        ref_imgs = [img.to(device, non_blocking=True) for img in ref_imgs]
        if args.gpu_augment and tgt_img.dtype == torch.uint8:
            from scsfm_hip import augment as hip_augment
            frames = torch.stack([tgt_img] + ref_imgs, dim=1).contiguous()  # [B, T, H, W, 3] uint8
            recs = hip_augment.draw_params(frames.shape[0], frames.shape[2], frames.shape[3])
            out = hip_augment.augment(frames, recs)                          # [T, B, 3, H, W] fp32
            tgt_img, ref_imgs = out[0], [out[i] for i in range(1, out.shape[0])]
            intrinsics = torch.from_numpy(hip_augment.update_intrinsics(intrinsics.numpy(), recs, frames.shape[3]))
        intrinsics = intrinsics.to(device, non_blocking=True).float()

        loss, loss_1, loss_2, loss_3 = train_step(args, disp_net, pose_net, optimizer, tgt_img, ref_imgs, intrinsics)

        # one host read-back per step instead of the reference's five .item() calls (train.py:277,290)
        vals = torch.stack([loss.detach(), loss_1.detach(), loss_2.detach(), loss_3.detach()]).tolist()
        if log_losses:
            train_writer.add_scalar('photometric_error', vals[1], n_iter)
            train_writer.add_scalar('disparity_smoothness_loss', vals[2], n_iter)
            train_writer.add_scalar('geometry_consistency_loss', vals[3], n_iter)
            train_writer.add_scalar('total_loss', vals[0], n_iter)
        losses.update(vals[0], args.batch_size)
        batch_time.update(time.time() - end)
        end = time.time()
        if is_main:
            with open(os.path.join(args.save_path, args.log_full), 'a') as csvfile:
                csv.writer(csvfile, delimiter='\t').writerow(vals)
            logger.train_bar.update(i + 1)
            if i % args.print_freq == 0:
                logger.train_writer.write('Train: Time {} Data {} Loss {}'.format(batch_time, data_time, losses))
        if i >= epoch_size - 1:
            break
        n_iter += 1
    return losses.avg[0]


@torch.no_grad()
def validate_without_gt(args, val_loader, disp_net, pose_net, epoch, logger, output_writers=[]):
    """Photometric / smoothness / geometry losses on the validation split, auto-mask off
    (train.py:302-362)."""
    global device
    batch_time = AverageMeter()
    losses = AverageMeter(i=4, precision=4)
    disp_net.eval()
    pose_net.eval()
    end = time.time()
    logger.valid_bar.update(0)
    for i, (tgt_img, ref_imgs, intrinsics, intrinsics_inv) in enumerate(val_loader):
        tgt_img = tgt_img.to(device)
        ref_imgs = [img.to(device) for img in ref_imgs]
        intrinsics = intrinsics.to(device).float()
        tgt_depth = [1 / disp_net(tgt_img)]
        ref_depths = [[1 / disp_net(ref_img)] for ref_img in ref_imgs]
        poses, poses_inv = compute_pose_with_inv(pose_net, tgt_img, ref_imgs)
        loss_1, loss_3 = compute_photo_and_geometry_loss(tgt_img, ref_imgs, intrinsics, tgt_depth, ref_depths, poses, poses_inv,
                                                         args.num_scales, args.with_ssim, args.with_mask, False, args.padding_mode)
        loss_2 = compute_smooth_loss(tgt_depth, tgt_img, ref_depths, ref_imgs)
        losses.update([float(loss_1), float(loss_1), float(loss_2), float(loss_3)])
        batch_time.update(time.time() - end)
        end = time.time()
        logger.valid_bar.update(i + 1)
        if i % args.print_freq == 0:
            logger.valid_writer.write('valid: Time {} Loss {}'.format(batch_time, losses))
    logger.valid_bar.update(len(val_loader))
    return losses.avg, ['Total loss', 'Photo loss', 'Smooth loss', 'Consistency loss']


@torch.no_grad()
def validate_with_gt(args, val_loader, disp_net, epoch, logger, output_writers=[]):
    """Depth metrics against ground truth; errors[1] (abs_rel) selects the best model
    (train.py:365-423)."""
    global device
    batch_time = AverageMeter()
    error_names = ['abs_diff', 'abs_rel', 'sq_rel', 'a1', 'a2', 'a3']
    errors = AverageMeter(i=len(error_names))
    disp_net.eval()
    end = time.time()
    logger.valid_bar.update(0)
    for i, (tgt_img, depth) in enumerate(val_loader):
        tgt_img = tgt_img.to(device)
        depth = depth.to(device)
        if depth.nelement() == 0:
            continue
        output_disp = disp_net(tgt_img)
        output_depth = 1 / output_disp[:, 0]
        if depth.nelement() != output_depth.nelement():
            b, h, w = depth.size()
            output_depth = torch.nn.functional.interpolate(output_depth.unsqueeze(1), [h, w]).squeeze(1)
        errors.update(compute_errors(depth, output_depth, args.dataset))
        batch_time.update(time.time() - end)
        end = time.time()
        logger.valid_bar.update(i + 1)
        if i % args.print_freq == 0:
            logger.valid_writer.write('valid: Time {} Abs Error {:.4f} ({:.4f})'.format(batch_time, errors.val[0], errors.avg[0]))
    logger.valid_bar.update(len(val_loader))
    return errors.avg, error_names


def compute_depth(disp_net, tgt_img, ref_imgs):
    tgt_depth = [1 / disp for disp in disp_net(tgt_img)]
    ref_depths = [[1 / disp for disp in disp_net(ref_img)] for ref_img in ref_imgs]
    return tgt_depth, ref_depths


def compute_pose_with_inv(pose_net, tgt_img, ref_imgs):
    poses, poses_inv = [], []
    for ref_img in ref_imgs:
        poses.append(pose_net(tgt_img, ref_img))
        poses_inv.append(pose_net(ref_img, tgt_img))
    return poses, poses_inv


if __name__ == '__main__':
    main()
