import os, sys, time, argparse
ROOT='/root/repo'; sys.path[:0]=[ROOT, ROOT+'/sc-sfmlearner-release_amd']
import torch
import bench
def run(tag, benchmark, channels_last):
    torch.backends.cudnn.benchmark = benchmark
    args = argparse.Namespace(batch=12, height=256, width=832, n_ref=2, dataset='kitti', e2e_steps=10, e2e_warmup=3)
    dev = torch.device('cuda:0')
    if channels_last:
        import models
        orig = models.DispResNet.__init__
    t0=time.time()
    r = bench.e2e_train(args, dev, 1, 0, torch.cuda.synchronize)
    print(tag, 'benchmark', benchmark, r['ms_per_step'], 'ms/step', r['train_images_per_sec'], 'img/s', 'wall %.1fs'%(time.time()-t0), flush=True)
run('default', False, False)
run('find', True, False)
