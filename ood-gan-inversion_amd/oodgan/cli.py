"""``python -m oodgan.cli --opt options/test/E4E_Face_test.yml`` — the harness around the accelerated path, with the
YAML option surface of the reference's run_ood_faceGAN_inversion.py:

    name, save_dir, directions_dir
    datasets: {<name>: {dataroot, editing: {direction, intensity}}}
    network_g: {type: ood_faceGAN_e4e, ...constructor kwargs...}
    path: {pretrain_network_g, param_key_g, strict_load_g}
    metrics: {psnr|ssim|lpips|identity: {crop_border, test_y_channel[, model_path]}}

Per image it does what the reference does (read -> [-1,1] RGB 1024² -> model -> save inversion + mask strip -> metrics)
and, when the build-defined block ``inversion: {wplus_steps: N, lr: 0.01, batch: B}`` (or ``--wplus-steps``) is
present, refines the encoder latents with N W+ Adam steps before the OOD forward (``streams: S``
in the same block advances the loop on S concurrent HIP streams, default 1; ``graph: true`` replays the plain forward from a
captured hipGraph).
``model_dict`` holds the reference's three variants (run_ood_faceGAN_inversion.py:23-27): the ``network_g`` blocks of
options/test/{E4E,ReStyle,FeatureStyle}_Face_test.yml resolve unchanged.  LPIPS / identity need third-party weights that
do not ship: they are reported as skipped."""
import argparse
import logging
import os
import time

import numpy as np
import torch
import yaml

from . import imgio
from .arch import ood_faceGAN_e4e, ood_faceGAN_FeatureStyle, ood_faceGAN_restyle
from .io import load_direction, load_network_g

model_dict = {                                   # run_ood_faceGAN_inversion.py:23-27
    'ood_faceGAN_e4e': ood_faceGAN_e4e,
    'ood_faceGAN_restyle': ood_faceGAN_restyle,
    'ood_faceGAN_FeatureStyle': ood_faceGAN_FeatureStyle,
}
IMG_EXT = ('.png', '.jpg', '.jpeg', '.bmp', '.webp')


def load_model(opts):
    """run_ood_faceGAN_inversion.py:30-47."""
    opt = dict(opts['network_g'])
    model_type = opt.pop('type')
    if model_type not in model_dict:
        raise KeyError(f'network_g.type {model_type!r} is not available in this build (have: {sorted(model_dict)})')
    model = model_dict[model_type](**opt)
    p = opts.get('path') or {}
    if p.get('pretrain_network_g'):
        load_network_g(model, p['pretrain_network_g'], p.get('param_key_g', 'params_ema'), p.get('strict_load_g', False))
    return model


def load_files_from_path(opt, directions_dir=None):
    """:49-62 — sorted by file name without its extension."""
    root = opt['dataroot']
    names = [n for n in os.listdir(root) if n.lower().endswith(IMG_EXT)]
    names = sorted(names, key=lambda x: x[:-4])
    return [os.path.join(root, n) for n in names], load_direction(directions_dir, opt.get('editing'))


def evaluate(gt_bgr, res_bgr, metrics, opt):
    """:89-126 — gt and result as [0,255] BGR arrays."""
    if metrics is None:
        metrics = {'psnr': [], 'ssim': [], 'lpips': [], 'identity': []}
    opt = opt or {}
    if opt.get('psnr'):
        metrics['psnr'].append(imgio.calculate_psnr(gt_bgr, res_bgr, crop_border=opt['psnr']['crop_border'],
                                                    test_y_channel=opt['psnr']['test_y_channel']))
    if opt.get('ssim'):
        metrics['ssim'].append(imgio.calculate_ssim(gt_bgr, res_bgr, crop_border=opt['ssim']['crop_border'],
                                                    test_y_channel=opt['ssim']['test_y_channel']))
    return metrics


def run(opts, wplus_steps=None, log=None):
    log = log or logging.getLogger('oodgan.cli')
    if not torch.cuda.is_available():
        raise RuntimeError('oodgan.cli needs a ROCm GPU: the HIP path has no CPU fallback')
    model = load_model(opts).cuda().eval()
    directions_dir = opts.get('directions_dir', './directions')
    save_root = os.path.join(opts.get('save_dir', './results'), opts['name'])
    inv = opts.get('inversion') or {}
    steps = int(wplus_steps if wplus_steps is not None else inv.get('wplus_steps', 0))
    lr = float(inv.get('lr', 0.01))
    streams = int(inv.get('streams', 1))
    lpips_weight = float(inv.get('lpips_weight', 0.0))       # opt-in perceptual term of the W+ loss (oodgan/lpips.py)
    lpips_state = torch.load(inv['lpips_path'], map_location='cpu') if inv.get('lpips_path') else None
    if lpips_weight and lpips_state is None:
        log.warning('inversion.lpips_weight without inversion.lpips_path: the LPIPS term runs on SEEDED AlexNet / lin weights (the lpips package\'s '
                    'pretrained weights are not part of this build)')
    graphed = None
    if inv.get('graph', False) and steps == 0:
        from .arch import GraphedForward
        graphed = GraphedForward(model)          # model(x) replayed from a hipGraph: -6 % latency per image at batch 1
    size = model.generator.size
    summary = {}
    for name, dopt in opts['datasets'].items():
        files, direction = load_files_from_path(dopt, directions_dir)
        save_dir = os.path.join(save_root, name)
        model.delta_latent.data += direction.cuda()
        times, metrics = [], None
        # ``inversion.batch`` files per call (default 1 = the reference's per-file loop, run_ood_faceGAN_inversion.py:158-182): images are
        # independent on this path, and the W+ loop of one image leaves most of the GPU idle (284 ms per image alone, 131 ms in a batch of 8)
        nb = max(1, int(inv.get('batch', 1)))
        for c0 in range(0, len(files), nb):
            chunk = files[c0:c0 + nb]
            bgrs = [imgio.imread(f).astype(np.float64) for f in chunk]
            x = torch.cat([imgio.image_to_input(bgr, size, device='cuda') for bgr in bgrs], 0)
            with torch.no_grad():
                t0 = time.time()
                if steps > 0:
                    out = model.invert(x, steps=steps, lr=lr, streams=streams, lpips_weight=lpips_weight, lpips_state=lpips_state)[0]
                else:
                    out = (graphed(x) if graphed is not None else model(x))[0]
                torch.cuda.synchronize()
                times += [(time.time() - t0) / len(chunk)] * len(chunk)
            for k, (f, bgr) in enumerate(zip(chunk, bgrs)):
                res = imgio.tensor2img(out[k:k + 1], rgb2bgr=True, min_max=(-1, 1))
                imgio.imwrite(os.path.join(save_dir, 'inversion', os.path.basename(f)), res)
                gt = bgr if bgr.shape[:2] == (size, size) else imgio.tensor2img(x[k:k + 1], rgb2bgr=True, min_max=(-1, 1)).astype(np.float64)
                metrics = evaluate(gt, res, metrics, opts.get('metrics'))
                masks = imgio.extract_masks(model.aligns, size, index=k)
                if masks is not None:
                    imgio.imwrite(os.path.join(save_dir, 'masks', os.path.basename(f)), masks)
        model.delta_latent.data -= direction.cuda()
        mean = lambda v: float(np.mean(v)) if v else float('nan')
        summary[name] = dict(n=len(files), time=mean(times), psnr=mean((metrics or {}).get('psnr')),
                             ssim=mean((metrics or {}).get('ssim')))
        log.info('Average process time of %s: %f', name, summary[name]['time'])
        log.info('Average PSNR of %s: %f', name, summary[name]['psnr'])
        log.info('Average SSIM of %s: %f', name, summary[name]['ssim'])
        for skipped in ('lpips', 'identity'):
            if (opts.get('metrics') or {}).get(skipped):
                log.info('%s of %s: skipped (third-party weights not available in this build)', skipped.upper(), name)
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--opt', default=None, required=True, help='the testing option file path')
    ap.add_argument('--wplus-steps', type=int, default=None, help='W+ refinement steps per image (default: inversion.wplus_steps or 0)')
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(name)s - %(levelname)s - %(message)s')
    with open(args.opt) as f:
        opts = yaml.load(f, Loader=yaml.FullLoader)
    return run(opts, args.wplus_steps)


if __name__ == '__main__':
    main()
