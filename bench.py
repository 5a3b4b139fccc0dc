#!/usr/bin/env python
"""Headline benchmark: train images/sec of the NAFNet-ref guided-restoration step
(BASELINE.json configs[1]: NAFNet-width32 + ref fusion, 512x512, sigma=15, bs=4/GPU).

    python bench.py --gpus N --steps K --warmup W
N>1: one rank per GPU, RCCL over xGMI.  Either the caller launches the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: WORLD_SIZE is set), or -- called
plainly, without WORLD_SIZE -- this script re-executes itself under torch.distributed.run with N ranks (the reference's
entry is `python -m torch.distributed.launch --nproc_per_node=N ...`, README.md:116).  Either way ONE JSON line with
n_gpus = the number of ranks that took part (`ranks_seen`, read back from the RCCL communicator).

A step = feed_train_data + optimize_parameters of RefGuidedImageCleanModel:
forward, L1, hand-written backward, gradient all-reduce (N>1), global-norm clip,
AdamW -- all HIP kernels, fp32, synthetic inputs resident in HBM.
Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work per image, SURVEY.md 8(d) (module-granular fp32 traffic; dense conv+matmul FLOPs, bwd = 2x fwd)
CFG2 = dict(B_alg=52.2e9, F_alg=2.00e12)
PEAK_HBM, PEAK_F32 = 8.0e12, 157.3e12
PEAK_BF16 = 2.5e15                    # dense bf16 MFMA (MI355X_MICROARCH.md); the split-bf16 kernels spend 6 products per fp32 product
PEAK_BX3 = PEAK_BF16 / 6.0


def make_opt(width, enc, batch_hw, dist_on, arch='nafnet', bucket_mb=64):
    if arch == 'restormer':      # BASELINE configs[2] / SURVEY 8d cfg3: Restormer-ref dim=nf=48
        net = dict(type='RestormerRefFusion', inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8],
                   num_refinement_blocks=4, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False,
                   LayerNorm_type='WithBias', dual_pixel_task=False, nf=48, ext_n_blocks=[4, 4, 4, 4],
                   reffusion_n_blocks=[2, 2, 2, 2])
    elif arch == 'drsformer':    # 007_drsformer_image_deraining_rain200l.yml network (DRSformer200L_SPA_RefFusion, no MEFC)
        net = dict(type='DRSformer200L_SPA_RefFusion', inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8],
                   heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', nf=48,
                   ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2])
    elif arch == 'drsformer_mefc':   # 008/009/010_drsformer_*.yml network (DRSformerRefFusion, with the MEFC sub-networks)
        net = dict(type='DRSformerRefFusion', inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8],
                   heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', nf=48,
                   ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2])
    elif arch == 'promptir':     # the reference's 001_promptir_all_in_one_restoration.yml network, with decoder=True (False raises: R4)
        net = dict(type='PromptIRRefFusion', inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8],
                   num_refinement_blocks=4, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False,
                   LayerNorm_type='WithBias', decoder=True, nf=48, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2])
    else:
        net = dict(type='NAFNetRefFusion', width=width, nf=width, enc_blk_nums=enc, dec_blk_nums=[1, 1, 1, 1],
                   middle_blk_num=1, ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 2])
    return {
        'model_type': 'RefGuidedImageCleanModel', 'num_gpu': 1, 'dist': dist_on, 'is_train': True,
        'network_g': net,
        'path': {},
        'train': {'optim_g': {'type': 'AdamW', 'lr': 2e-4, 'ref_lr': 1e-4, 'weight_decay': 1e-4, 'betas': [0.9, 0.999]},
                  'scheduler': {'type': 'CosineAnnealingRestartCyclicLR', 'periods': [306000, 694000],
                                'restart_weights': [1, 1], 'eta_mins': [3e-4, 1e-6]},
                  'pixel_opt': {'type': 'L1Loss', 'loss_weight': 1, 'reduction': 'mean'},
                  'use_grad_clip': True, 'total_iter': 1000000, 'warmup_iter': -1},
        'logger': {'check_freq': 10 ** 9}, 'val': {}, 'scale': 1, 'dist_bucket_mb': bucket_mb,
    }


def _cpu_baseline_worker(width, enc, H, batch, budget):
    """child process: the CPU oracle (a port of the reference's path, pinned to the reference by tests/golden) timed on
    the metric's own workload: train steps of the same network on `batch` x H x H pairs.  The thread count is chosen by a
    short scaling probe (steps at 128x128 under 8 / 16 / 32 / 64 threads): torch's MKLDNN fp32 convolutions stop scaling
    -- and can collapse -- far below the 256 hardware threads of the GPU hosts.  Prints one JSON object."""
    from oracle import nafnet_ref_oracle as O
    cfg = O.default_cfg(width=width, nf=width, enc_blk_nums=enc, ext_n_blocks=[4, 4, 4, 4],
                        reffusion_n_blocks=[2, 2, 2, 2, 2])
    tr = O.OracleTrainer(O.synth_params(cfg, seed=0), cfg)
    lq, gt, ref = O.synth_pair(1, 128, 128, seed=1)
    probe = {}
    for t in [t for t in (8, 16, 32, 64) if t <= (os.cpu_count() or 1)] or [1]:
        torch.set_num_threads(t)
        tr.step(lq, gt, ref)                              # thread-pool / allocator warm-up at this width
        t0 = time.time()
        tr.step(lq, gt, ref)
        probe[t] = time.time() - t0
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    lq, gt, ref = O.synth_pair(batch, H, H, seed=2)
    tr.step(lq, gt, ref)                                  # 1 warm-up step at the sample's shape
    t0 = time.time()
    n = 0
    while True:
        tr.step(lq, gt, ref)
        n += 1
        if time.time() - t0 > budget or n >= 4:
            break
    print(json.dumps({'n': n, 'dt': time.time() - t0, 'threads': threads, 'probe': {str(k): round(v, 3) for k, v in probe.items()}}))


def cpu_baseline(width, enc, size, batch, budget=20.0, hard_timeout=480.0):
    """Reported baseline only (never the thing shipped): `oracle/` timed on this host's cores in a child process with a
    hard timeout, on a bounded sample of the metric's workload: whole train steps of the same network on the metric's OWN per-GPU
    batch (`batch` x size x size pairs; round 6 -- rounds 1-5 timed one pair), 1 warm-up + 1..4 timed steps (about 20 - 40 s)."""
    import subprocess
    code = (f'import sys; sys.path.insert(0, {ROOT!r}); import bench; '
            f'bench._cpu_baseline_worker({width}, {enc!r}, {size}, {batch}, {budget})')
    try:
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=hard_timeout,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES=''))
        r = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001  (timeout / crash of the baseline must not lose the GPU measurement)
        return {'value': None, 'unit': 'images/sec', 'cores': None, 'kind': 'port', 'sample': f'failed: {type(e).__name__}'}
    return {'value': r['n'] * batch / r['dt'], 'unit': 'images/sec', 'cores': r['threads'], 'kind': 'port',
            'host_cpus': os.cpu_count(),
            'sample': f"1 warm-up + {r['n']} timed train step(s) of the same network on the metric's own per-GPU batch ({batch} x {size}x{size} "
                      f"pairs) in {r['dt']:.1f} s, torch-CPU fp32 oracle (oracle/nafnet_ref_oracle.py); threads chosen by a scaling probe "
                      f"(seconds per 128x128 step by thread count: {r['probe']})"}


def _child_bench(args, extra, env=None, timeout=600):
    """this script again in a child process (1 GPU, no secondary legs); returns its parsed JSON line"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--arch', args.arch, '--batch', str(args.batch), '--size',
           str(args.size), '--width', str(args.width), '--enc', args.enc, '--no-cpu-baseline', '--no-roofline', '--no-f32-exact',
           '--no-matcher-active'] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])


def f32_exact_run(args):
    """the same workload on the exact-fp32 MFMA path (TDR_MATH=f32, v_mfma_f32_32x32x2_f32: bitwise an fmaf chain), a few
    steps in a child process: the price of the split arithmetic's speed-up is visible next to the headline number"""
    try:
        r = _child_bench(args, ['--steps', '10', '--warmup', '2'], env={'TDR_MATH': 'f32'})
        out = {'ms_per_step': r['ms_per_step'], 'value': r['value'], 'unit': r['unit'], 'steps': r['steps'], 'dtype': r['dtype'],
               'arithmetic': 'exact fp32 MFMA (TDR_MATH=f32), same workload, same code path otherwise'}
        if 'roofline_step' in r:
            out['roofline_step'] = r['roofline_step']
        return out
    except Exception as e:  # noqa: BLE001
        return {'ms_per_step': None, 'note': f'failed: {type(e).__name__}'}


def fast_mode_run(args):
    """DISCLOSED FAST MODE, not the headline: the same workload on the opt-in 2-way fp16 split (TDR_MATH=hx2: 22-bit operand
    significands inside the fp16 window, loss-scaled backward with a device-resident step guard, fp16 pair planes in the MASA
    encoder) -- narrower than the reference's fp32 arithmetic, reported next to the reference-arithmetic `value` for what it buys"""
    try:
        r = _child_bench(args, ['--steps', '20', '--warmup', '3'], env={'TDR_MATH': 'hx2'})
        out = {'ms_per_step': r['ms_per_step'], 'value': r['value'], 'unit': r['unit'], 'steps': r['steps'], 'dtype': r['dtype'],
               'guard': r.get('guard'),
               'arithmetic': 'opt-in TDR_MATH=hx2: 2-way fp16 split (3 f16 MFMA products per fp32 product, fp32 accumulate), loss-scaled backward; '
                             'NARROWER than the reference (22-bit operands, fp16 exponent window, a guard that may skip steps)'}
        if 'roofline_step' in r:
            out['roofline_step'] = r['roofline_step']
        return out
    except Exception as e:  # noqa: BLE001
        return {'ms_per_step': None, 'note': f'failed: {type(e).__name__}'}


def matcher_active_run(args, ref_size=640):
    """the same step with a reference LARGER than lq (the situation of the shipped YAMLs: 384x384 crops against 512x512
    generated references, image_restoration_ref_model.py:219-247): the frozen DINOv2 ViT-B/14 window matcher runs every step"""
    try:
        r = _child_bench(args, ['--steps', '5', '--warmup', '2', '--dino-ref-size', str(ref_size)])
        out = {'ms_per_step': r['ms_per_step'], 'value': r['value'], 'unit': r['unit'], 'steps': r['steps'],
               'dino_match': r['config']['dino_match'] + '; matcher arithmetic = the step\'s fp32-faithful split (default since round 4)',
               'guard': r.get('guard')}
        try:      # the opt-in single-product matcher (TDR_DINO_MATH=h1: arg-max pinned on random-init weights only)
            h = _child_bench(args, ['--steps', '5', '--warmup', '2', '--dino-ref-size', str(ref_size)], env={'TDR_DINO_MATH': 'h1'})
            out['h1_matcher_opt_in'] = {'ms_per_step': h['ms_per_step'], 'value': h['value']}
        except Exception:  # noqa: BLE001
            pass
        return out
    except Exception as e:  # noqa: BLE001
        return {'ms_per_step': None, 'note': f'failed: {type(e).__name__}'}


def bench_i2t(a, world, rank, local):
    """BASELINE configs[3]: the stage-A image-to-text mapping train step of
    scripts/train/main_train_i2t_mapping.py:704-760 -- frozen CLIP ViT-H/14 (1280 / 32 layers / 16 heads / MLP 5120) on the batch
    resized to 224x224, Mapper(1280 -> 1024, 20 words) forward + backward, the injected cross-attention at the four SD shapes with
    trainable to_k_global / to_v_global, MSE, clip_grad_norm_ 1.0, AdamW; the absent third-party SD UNet / VAE / text transformer are
    the fixed random linear stand-in SURVEY prescribes (textualdegremoval_amd/stage_a.py).  Random-init weights, synthetic batch."""
    from textualdegremoval_amd import kernels as K
    from textualdegremoval_amd import stage_a as SA
    from textualdegremoval_amd.clip_vision import random_clip_state_dict
    vit = {'H': (1280, 5120, 32, 16, 'gelu', 0.334e12), 'L': (1024, 4096, 24, 16, 'quick_gelu', 0.1626e12)}[a.clip]
    torch.manual_seed(0)
    # --arch tr: the textual-restoration mapping step (main_train_tr_mapping.py:757-812) -- frozen Mapper, trained CleanMapper (with the
    # optimiser over the CleanMapper: the script as written updates nothing)
    cls = SA.TRMappingTrainer if a.arch == 'tr' else SA.I2TMappingTrainer
    tr = cls(random_clip_state_dict(vit[0], vit[1], vit[2]), vit[3], SA.stage_a_stub(seed=0), clip_act=vit[4],
             num_words=20, lr=1e-4 * a.batch * world, dist_on=world > 1, bucket_mb=a.bucket_mb)
    batch = {k: v.cuda() for k, v in SA.synthetic_batch(a.batch, size=a.size, seed=rank).items()}
    for _ in range(4 + a.warmup):
        tr.step(batch)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.step(batch)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    comm = tr.reducer.comm
    if world > 1 and a.backend == 'nccl' and comm is None:
        sys.exit('bench.py: backend nccl but the RCCL data plane (tdr_comm_*) is not in use')
    if rank != 0:
        return
    g = tr.optimizer.guard.read()
    n_map = sum(p.numel() for k, p in zip(tr.names, tr.params) if k.startswith('mapping_'))
    tokens = 257
    # dense FLOPs per image: CLIP forward + Mapper fwd + 2x bwd (2 * tokens * params of the patch MLPs; the class-token
    # MLPs see one token) + the cross-attention projections / products of the four levels (fwd + the dk / dv side of the backward)
    f_mapper = 3 * 2 * (tokens - 1 + 1) * (n_map / 2)
    ips = world * a.batch * a.steps / dt
    line = {'metric': f'train images/sec (stage-A {"TR" if a.arch == "tr" else "I2T"} mapping, {a.size}x{a.size}, bs={a.batch}/GPU)', 'value': ips, 'unit': 'images/sec',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'hx2': 'f32 (2xfp16-split MFMA forward, 3xbf16-split MFMA backward; fp32 accumulate)', 'bx3': 'f32 (3xbf16-split MFMA)',
                      'f32': 'f32 (exact fp32 MFMA)', 'h1': 'f16 (single fp16 MFMA product)'}[K.MATH],
            'data': 'synthetic',
            'guard': {'skipped_total': int(g.skipped), 'applied_steps': int(g.step)},
            'config': {'workload': ('textual-restoration mapping step (main_train_tr_mapping.py:757-812): the stage-A pipeline with the Mapper and its to_k / '
                                    'to_v FROZEN and a CleanMapper(1024->1024, 20 words) trained (optimiser over the CleanMapper: DESIGN R9); ' if a.arch == 'tr' else '') +
                                   f'BASELINE configs[3]: I2T-mapper training step, CLIP ViT-{a.clip}/14 frozen + Mapper(1280->1024, 20 words) + injected '
                                   f'cross-attention at 4096/1024/256/64 tokens (to_k_global/to_v_global trained), MSE, clip 1.0, AdamW; SD UNet/VAE/text '
                                   f'transformer = fixed random linear stand-in (SURVEY 8d cfg4), {a.size}x{a.size}, bs={a.batch}/GPU',
                       'global_batch': world * a.batch, 'parallelism': f'dp{world}', 'trained_parameters': sum(p.numel() for p in tr.params),
                       'collectives': ('none (1 GPU)' if world == 1 else
                                       ('tdr_comm_* (RCCL through the C ABI)' if comm is not None else f'torch.distributed ({a.backend})')),
                       'hip_graph': bool(tr.use_hip_graph)},
            'final_loss': float(loss.item()),
            'roofline_step': {'alg_flop_per_image': vit[5] + f_mapper, 'clip_fwd_flop_per_image': vit[5], 'mapper_fwd_bwd_flop_per_image': f_mapper,
                              'achieved_tflops': (vit[5] + f_mapper) * ips / world / 1e12,
                              'achieved_split_flop_frac': (vit[5] + f_mapper) * ips / world / (PEAK_BF16 / 3.0),
                              'weight_bytes_touched_per_step': 28.0 * sum(p.numel() for p in tr.params),
                              'note': 'fp32-equivalent FLOPs against the 2-way-split ceiling (833 TF); the Mapper also streams 28 B per '
                                      'trained parameter per step (weights fwd + bwd, gradients, AdamW state)'}}
    print(json.dumps(line), flush=True)


def pmc_traffic(prefixes):
    """HBM bytes per launch of a kernel family (kernel-name prefixes) from the committed PMC passes (profiles/pmc_collect.sh ->
    profiles/r<N>/pmc_traffic.json).  The file records the hashes of the kernel sources it was collected on: a mismatch means the
    counters describe older kernels and the figure is reported as stale (null) instead of being passed off as current."""
    import hashlib
    srcs = {'tdr_conv_bx3_sha256': 'tdr_conv_bx3.hip', 'tdr_conv_p16_sha256': 'tdr_conv_p16.hip'}
    pmc = path = None
    for rnd in ('r6', 'r5', 'r4', 'r3', 'r2'):     # newest collection first
        cand = os.path.join(ROOT, 'profiles', rnd, 'pmc_traffic.json')
        try:
            with open(cand) as fh:
                pmc, path = json.load(fh), f'profiles/{rnd}/pmc_traffic.json'
            break
        except (OSError, ValueError):
            continue
    if pmc is None:
        return None, 'no PMC file (profiles/pmc_collect.sh not run)'
    for key, fn in srcs.items():
        try:
            with open(os.path.join(ROOT, 'textualdegremoval_amd', 'csrc', fn), 'rb') as fh:
                cur = hashlib.sha256(fh.read()).hexdigest()
        except OSError:
            cur = None
        if pmc.get('meta', {}).get(key) != cur:
            return None, f'stale: {path} was collected on another revision of csrc/{fn}'
    tot, n = 0.0, 0
    for name, v in pmc['kernels'].items():
        if any(name.startswith(p) for p in prefixes):
            tot += (v['read_bytes_per_launch'] + v['write_bytes_per_launch']) * v['launches']
            n += v['launches']
    if not n:
        return None, 'kernel family not in the PMC file'
    return tot / n, ('bytes/launch of the 3x3 stride-1 forward + data-gradient family (FETCH_SIZE + WRITE_SIZE in separate passes, calibrated '
                     f'on copies of known size, {path}; collected on this revision of the kernel sources)')


def pmc_clock(prefix):
    """effective shader clock and matrix-pipe busy share (in CYCLES) of a kernel family under its own load, from the committed counter pass
    (profiles/pmc_clock.sh -> profiles/r<N>/pmc_clock.json: GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, dispatch timestamps of the same pass).
    The roofline peak is the guide's 2.4 GHz figure; the chip clocks MFMA-dense kernels at 1.8 - 2.0 GHz (MI355X_MICROARCH.md, DVFS give-back), so
    `frac` understates how busy the matrix pipes are per cycle.  Context only -- `frac` stays priced against the nominal peak."""
    for rnd in ('r6',):
        try:
            with open(os.path.join(ROOT, 'profiles', rnd, 'pmc_clock.json')) as fh:
                js = json.load(fh)
        except (OSError, ValueError):
            continue
        rows = [k for k in js.get('kernels', []) if k['kernel'].startswith(prefix)]
        w = sum(k['launches'] * k['avg_us'] for k in rows)
        if not w:
            return None
        return {'effective_ghz': sum(k['effective_ghz'] * k['launches'] * k['avg_us'] for k in rows) / w,
                'mfma_busy_cycle_frac': sum(k['mfma_busy_cycle_frac'] * k['launches'] * k['avg_us'] for k in rows) / w,
                'by_instantiation': [{'kernel': k['kernel'], 'grid': k['grid'], 'avg_us': round(k['avg_us'], 1), 'effective_ghz': round(k['effective_ghz'], 3),
                                      'mfma_busy_cycle_frac': round(k['mfma_busy_cycle_frac'], 3)} for k in rows],
                'source': f'profiles/{rnd}/pmc_clock.json (time-weighted over the instantiations; kernels serialised by the counter pass; nominal peak = 2.4 GHz)'}
    return None


def pmc_step_bytes():
    """measured HBM bytes of one whole train step from the same PMC collection (sum over every kernel of the step), or None"""
    for rnd in ('r6', 'r5', 'r4', 'r3'):
        try:
            with open(os.path.join(ROOT, 'profiles', rnd, 'pmc_traffic.json')) as fh:
                pmc = json.load(fh)
        except (OSError, ValueError):
            continue
        tot = pmc.get('meta', {}).get('step_total_bytes')
        if tot:
            return float(tot), f'profiles/{rnd}/pmc_traffic.json'
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--clip', default='H', choices=['H', 'L'], help='--arch i2t: CLIP ViT-H/14 (SD-2.1, the Mapper input width 1280) or ViT-L/14 geometry')
    ap.add_argument('--arch', default='nafnet', choices=['nafnet', 'restormer', 'promptir', 'drsformer', 'drsformer_mefc', 'i2t', 'tr'],
                    help="nafnet: the headline workload (BASELINE configs[1]); restormer: configs[2]'s per-GPU workload "
                         '(Restormer-ref dim 48, 256x256, bs 8) -- a secondary measurement, not the metric line')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--width', type=int, default=32)
    ap.add_argument('--enc', type=str, default='1,1,1,28')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-f32-exact', action='store_true', help='skip the exact-fp32 (TDR_MATH=f32) comparison run')
    ap.add_argument('--no-matcher-active', action='store_true', help='skip the DINOv2-matcher-active comparison run (ref 640x640)')
    ap.add_argument('--dino-ref-size', type=int, default=0,
                    help='feed a reference image of this size (> --size): the frozen DINOv2 ViT-B/14 window matcher runs every step '
                         '(random-init weights); 0 = ref of the lq size, where the match is the identity')
    ap.add_argument('--bucket-mb', type=float, default=64.0, help='gradient all-reduce bucket size (MiB) of the data-parallel step')
    ap.add_argument('--backend', default='nccl', help='nccl (= RCCL over xGMI; default) | gloo (multi-process smoke test on one GPU)')
    ap.add_argument('--pg-timeout', type=float, default=120.0, help='seconds the torch.distributed rendezvous may take')
    ap.add_argument('--rccl-dry-run', action='store_true',
                    help='with --gpus N: bring the RCCL data plane up through the C ABI (tdr_comm_*), verify a broadcast and an all-reduce, time '
                         'one 64 MiB exchange, print ONE JSON line and exit -- or fail loudly with the name of the bring-up stage that did')
    ap.add_argument('--math', default=None, choices=['bx3', 'f32', 'hx2', 'h1'],
                    help='arithmetic of the dense contractions (default: the library default bx3 -- except the BASELINE configs[4] workload, '
                         '`--arch restormer --size 512 --batch 2`, whose config names fp16 MFMA: h1)')
    a = ap.parse_args()
    if a.batch is None:
        a.batch = 8 if a.arch in ('restormer', 'promptir', 'drsformer', 'drsformer_mefc') else 4
    if a.size is None:
        a.size = {'restormer': 256, 'promptir': 384, 'drsformer': 256, 'drsformer_mefc': 256}.get(a.arch, 512)
    enc = [int(v) for v in a.enc.split(',')]
    if a.math is None and (a.arch, a.size, a.batch) == ('restormer', 512, 2) and 'TDR_MATH' not in os.environ:
        a.math = 'h1'                  # BASELINE configs[4]: "Restormer-ref 512x512 bs=2/GPU ... fp16 MFMA"
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # called plainly: become the launcher (one rank per GPU; rank 0's JSON line and every rank's stderr pass through)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr',
               '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')).returncode)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        sys.exit(f'bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks')
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local % torch.cuda.device_count())
    one_rank_dry = a.rccl_dry_run and world == 1 and os.environ.get('TDR_FORCE_COLLECTIVES') == '1'
    if one_rank_dry:                    # a 1-GPU box exercising the same calls (tests/test_hip_dp_smoke.py)
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if world > 1 or one_rank_dry:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.backend == 'nccl':
            # the N-GPU line is an RCCL-through-the-C-ABI measurement or it is an error: no silent torch.distributed fallback
            os.environ.setdefault('TDR_COMM', 'rccl')
        import datetime
        try:
            dist.init_process_group(a.backend, timeout=datetime.timedelta(seconds=float(a.pg_timeout)))
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f'bench.py: rank {rank}/{world}: torch.distributed rendezvous failed ({type(e).__name__}: {e})\n')
            sys.exit(2)
    try:
        return _main_body(a, world, rank, local, enc)
    except Exception as e:  # noqa: BLE001
        from textualdegremoval_amd.parallel import DataPlaneUnavailable
        if isinstance(e, DataPlaneUnavailable):
            # one line, non-zero exit: the N-GPU line is an RCCL measurement or an error (the bring-up watchdog covers ranks that hang)
            sys.stderr.write(f'bench.py: rank {rank}/{world}: RCCL data plane unavailable: {e}\n')
            sys.exit(2)
        raise


def rccl_dry_run(a, world, rank):
    """The first thing to run on a new multi-GPU node: every stage of the RCCL bring-up and one exchange of each kind, each stage named in
    the failure message (reference: models/base_model.py:76-82 wraps the network in DistributedDataParallel; this is what stands in its
    place).  Exit code 2 + one stderr line naming the stage on failure; one JSON line on success."""
    from textualdegremoval_amd import parallel
    stages, t_stage = [], time.perf_counter()

    def done(name, **extra):
        nonlocal t_stage
        now = time.perf_counter()
        stages.append(dict(stage=name, ok=True, ms=(now - t_stage) * 1e3, **extra))
        t_stage = now

    def fail(name, why):
        sys.stderr.write(f"bench.py --rccl-dry-run: rank {rank}/{world}: FAILED at stage '{name}': {why}\n")
        sys.stderr.flush()
        if dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
        sys.exit(2)
    done('torch.distributed rendezvous (side channel)', backend=dist.get_backend())
    os.environ['TDR_COMM'] = 'rccl'                       # strict: a failed bring-up raises instead of falling back to torch collectives
    try:
        comm = parallel.data_plane()
    except parallel.DataPlaneUnavailable as e:
        fail('RCCL communicator bring-up (librccl resolves -> ncclGetUniqueId -> broadcast of the id -> ncclCommInitRank -> agreement)', str(e))
    if comm is None:
        fail('RCCL communicator bring-up', 'no communicator was created (world size 1 without TDR_FORCE_COLLECTIVES=1, or backend != nccl)')
    from textualdegremoval_amd import _lib
    seen = int(_lib.load().tdr_comm_world(comm.handle))
    if seen != world:
        fail('communicator size', f'tdr_comm_world = {seen}, launcher world = {world}')
    done('tdr_comm_init (ncclCommInitRank through the C ABI)', ranks_seen=seen)
    try:
        t = torch.full((1 << 16,), float(rank + 1), device='cuda')
        comm.broadcast(t, root=0)
        torch.cuda.synchronize()
        if not bool((t == 1.0).all()):
            fail('tdr_comm_broadcast', f'rank {rank} holds {t[0].item()} after a broadcast of 1.0 from rank 0')
        done('tdr_comm_broadcast (64 Ki floats from rank 0)')
        t = torch.arange(1 << 16, device='cuda', dtype=torch.float32) * float(rank + 1)
        comm.allreduce(t, average=True)
        torch.cuda.synchronize()
        want = torch.arange(1 << 16, device='cuda', dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        err = float((t - want).abs().max())
        if err > 1e-3 * float(want.abs().max()) / 1e3:
            fail('tdr_comm_allreduce (mean)', f'max |got - expected| = {err}')
        done('tdr_comm_allreduce (mean of rank-dependent data, checked element-wise)', max_abs_err=err)
        big = torch.ones(16 << 20, device='cuda')         # 64 MiB: one gradient bucket
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        comm.allreduce(big, average=True, stream=side.cuda_stream)      # warm-up (ring / channel set-up)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(5):
            comm.allreduce(big, average=True, stream=side.cuda_stream)
        e1.record(side)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        if not bool((big == 1.0).all()):
            fail('64 MiB bucket all-reduce on the comm stream', 'the averaged ones are not ones')
        done('64 MiB bucket all-reduce on a second HIP stream (x5)', ms_per_exchange=ms,
             bus_GBps=(2.0 * (world - 1) / max(world, 1)) * big.numel() * 4 / (ms * 1e-3) / 1e9 if world > 1 else None)
    except SystemExit:
        raise
    except Exception as e:  # noqa: BLE001
        fail(f'after "{stages[-1]["stage"]}"', f'{type(e).__name__}: {e}')
    allst = [None] * world
    dist.all_gather_object(allst, stages)
    if rank == 0:
        print(json.dumps({'rccl_dry_run': 'ok', 'n_gpus': world, 'stages': stages,
                          'slowest_rank_ms_by_stage': [max(r[i]['ms'] for r in allst) for i in range(len(stages))]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def _main_body(a, world, rank, local, enc):
    if a.rccl_dry_run:
        if not dist.is_initialized():
            sys.exit('bench.py --rccl-dry-run needs --gpus N > 1 (or TDR_FORCE_COLLECTIVES=1 under a one-rank launcher)')
        return rccl_dry_run(a, world, rank)
    if a.arch in ('i2t', 'tr'):
        bench_i2t(a, world, rank, local)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    from textualdegremoval_amd.models import create_model
    from textualdegremoval_amd.utils.synthetic import randomize_gates, synthetic_pair
    from textualdegremoval_amd import kernels as K
    if a.math is not None:
        K.set_math(a.math)

    torch.manual_seed(0)                                   # identical initial weights on every rank
    opt = make_opt(a.width, enc, a.size, world > 1, a.arch, bucket_mb=a.bucket_mb)
    if a.dino_ref_size > a.size:
        from textualdegremoval_amd.dino import random_vit_b14_state_dict
        ck = f'/tmp/tdr_dino_vitb14_rank{rank}.pth'
        torch.save(random_vit_b14_state_dict(seed=0), ck)
        opt['path']['pretrain_dino'] = ck
    model = create_model(opt)
    randomize_gates(model.net_g)
    data = synthetic_pair(a.batch, a.size, a.size, seed=1234 + rank)
    if a.dino_ref_size > a.size:                           # clean image of the larger size; lq/gt = a crop of it (+ noise)
        big = synthetic_pair(a.batch, a.dino_ref_size, a.dino_ref_size, seed=1234 + rank)
        o = (a.dino_ref_size - a.size) // 2 // max(a.size // 4, 1) * max(a.size // 4, 1)
        data['ref'] = big['gt']
        data['gt'] = big['gt'][:, :, o:o + a.size, o:o + a.size].contiguous()
        data['lq'] = (data['gt'] + (big['lq'] - big['gt'])[:, :, o:o + a.size, o:o + a.size]).contiguous()
    data = {k: v.cuda() for k, v in data.items()}          # inputs resident in HBM before the timed region

    def step(it):
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_train_data(data)
        model.optimize_parameters(it)

    it = 0
    # one-time set-up, like building the model: the first steps of a shape run eagerly (workspaces, gradient-arena
    # layout, weight-pack plan) and then the step is captured into hipGraphs; none of that is steady-state work.
    # (step 1 also carries the fp16-window survey; the capture happens on the first step after two eager ones)
    for _ in range(5 if getattr(model, 'use_hip_graph', True) and os.environ.get('TDR_GRAPH', '1') == '1' else 1):
        it += 1
        step(it)
    for _ in range(a.warmup):
        it += 1
        step(it)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def guard_state():
        g = getattr(getattr(model, 'optimizer_g', None), 'guard', None)
        return g.read() if g is not None else None

    g_before = guard_state()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        it += 1
        step(it)
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = model.get_current_log()['l_pix']
    g_after = guard_state()
    red = getattr(model, 'grad_reducer', None)
    comm = getattr(red, 'comm', None)
    if world > 1:
        from textualdegremoval_amd import _lib
        ranks_seen = _lib.load().tdr_comm_world(comm.handle) if comm is not None else dist.get_world_size()
        if a.backend == 'nccl' and comm is None:
            sys.exit('bench.py: backend nccl but the RCCL data plane (tdr_comm_*) is not in use -- refusing to report an N-GPU line')
    else:
        ranks_seen = 1

    # ---- N > 1: what the first real multi-GPU run needs to diagnose itself (it runs on a node this repo's builder never touches):
    # per-rank view of the job, the gradient exchange timed with HIP events on the comm stream, and cross-rank agreement checks
    scale_diag = None
    if world > 1:
        import hashlib
        import socket
        red.timing = []
        for _ in range(3):                                # instrumented steps (outside the timed region)
            it += 1
            step(it)
        barrier()
        tsum = red.comm_timing_summary() if hasattr(red, 'comm_timing_summary') else {}
        red.timing = None
        flat = getattr(red, 'flat', None)
        gsum = float(flat.double().sum().item()) if flat is not None else None
        gabs = float(flat.double().abs().sum().item()) if flat is not None else None
        ghash = hashlib.sha1(flat.cpu().numpy().tobytes()).hexdigest()[:16] if flat is not None else None
        psum = float(sum(p.detach().double().sum().item() for p in model.net_g.parameters()))
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        mine = {'rank': rank, 'local_rank': local, 'host': socket.gethostname(), 'device': torch.cuda.current_device(),
                'gpu': props.name, 'ranks_seen': int(ranks_seen), 'buckets': len(getattr(red, 'buckets', [])),
                'arena_MB': (flat.numel() * 4 / 1e6) if flat is not None else None,
                'ms_per_step_local': dt_local / a.steps * 1e3, 'exchange': tsum,
                'grad_sum_after_allreduce': gsum, 'grad_abs_sum_after_allreduce': gabs, 'grad_sha1_16': ghash, 'param_sum': psum,
                'hbm_peak_allocated_gb': torch.cuda.max_memory_allocated() / 1e9}
        # ---- what the exchange costs on the wall clock: the same captured step with every collective left out (dry_exchange), timed the
        # same way.  Last thing this process does with the model: the replicas diverge from here on.
        barrier()
        red.dry_exchange = True
        for _ in range(2):
            it += 1
            step(it)
        barrier()
        t1 = time.perf_counter()
        n_dry = max(5, min(a.steps, 10))
        for _ in range(n_dry):
            it += 1
            step(it)
        torch.cuda.synchronize()
        mine['compute_only_ms_per_step'] = (time.perf_counter() - t1) / n_dry * 1e3
        red.dry_exchange = False
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        if rank == 0:
            t_meas = dt / a.steps * 1e3
            t_comp = max(r['compute_only_ms_per_step'] for r in allr)
            waited = max((r['exchange'].get('compute_stream_waited_ms') or 0.0) / 3.0 for r in allr)      # 3 instrumented steps
            pred = t_comp + waited
            scale_summary = {
                'ms_per_step': t_meas, 'compute_only_ms_per_step_slowest_rank': t_comp,
                'compute_only_ms_per_step_by_rank': [r['compute_only_ms_per_step'] for r in allr],
                'exposed_exchange_ms_per_step': t_meas - t_comp,
                'exposed_exchange_ms_per_step_by_events': waited,
                'predicted_ms_per_step': pred, 'predicted_over_measured': pred / t_meas,
                'efficiency_vs_own_compute': t_comp / t_meas,
                'checks': {'prediction_within_10pct': abs(pred - t_meas) <= 0.10 * t_meas,
                           'ranks_within_5pct_of_each_other': (max(r['ms_per_step_local'] for r in allr) <=
                                                               1.05 * min(r['ms_per_step_local'] for r in allr)),
                           'all_ranks_see_world': all(r['ranks_seen'] == world for r in allr)},
                'note': 'compute_only = the captured data-parallel step replayed with every collective skipped (same graphs and stream '
                        'dependencies); exposed = measured - compute_only; by_events = what the compute stream waited for the comm stream in 3 '
                        'instrumented steps.  The driver computes scaling efficiency itself from the per-N values; this block only says where '
                        'a shortfall comes from (compute imbalance between ranks vs exposed exchange time)'}
            scale_diag = {'summary': scale_summary, 'per_rank': allr,
                          'gradients_identical_across_ranks': len({r['grad_sha1_16'] for r in allr}) == 1,
                          'parameters_identical_across_ranks': len({r['param_sum'] for r in allr}) == 1,
                          'all_ranks_see_world': all(r['ranks_seen'] == world for r in allr),
                          'slowest_rank_ms_per_step': max(r['ms_per_step_local'] for r in allr),
                          'fastest_rank_ms_per_step': min(r['ms_per_step_local'] for r in allr),
                          'note': 'exchange.*: HIP events on the comm stream around every bucket all-reduce of 3 extra steps; compute_stream_waited_ms '
                                  'is what the optimiser graph had to wait for after the last backward segment (the exposed part)'}

    # ---- roofline of the dominant kernel family (3x3 stride-1 implicit GEMM on the fp32 matrix
    # cores: masa_enc forward + data-gradient launches), measured with HIP events on the launch stream
    # (every rank runs the instrumented step -- it contains the gradient all-reduce -- but only rank 0 reports)
    roof, roof_other = None, []
    if not a.no_roofline:
        recs = []          # (flop, e0, e1, alg bytes, family key)
        orig = K.conv_forward
        orig_p16, orig_wg, orig_wg16, orig_grp = K.conv3x3_p16, K.conv_wgrad, K.wgrad3x3_p16, K.wgrad1x1_group

        def _ev():
            return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(x, wp, Mpad, Cout, KH, stride=1, dil=1, pad=0, **kw):
            if KH == 3 and stride == 1 and dil == 1 and kw.get('wp_ns', 0) == 0:
                e0, e1 = _ev()
                e0.record()
                out = orig(x, wp, Mpad, Cout, KH, stride=stride, dil=dil, pad=pad, **kw)
                e1.record()
                recs.append((2.0 * x.shape[0] * Cout * x.shape[1] * 9 * out.shape[2] * out.shape[3], e0, e1,
                             4.0 * (x.numel() + out.numel()), ('conv', getattr(wp, 'fmt', 0))))
                return out
            return orig(x, wp, Mpad, Cout, KH, stride=stride, dil=dil, pad=pad, **kw)

        def timed_p16(x16, wp, Mpad, Cout, **kw):
            e0, e1 = _ev()
            e0.record()
            out = orig_p16(x16, wp, Mpad, Cout, **kw)
            e1.record()
            px = x16.N * x16.H * x16.W
            nout = (1 if kw.get('want32', True) else 0) + (1 if kw.get('want16', False) else 0)
            # (algorithmic bytes = the fp32 tensors of SURVEY 8d: 4 B per element; triple planes move 6)
            recs.append((2.0 * px * Cout * x16.C * 9, e0, e1, 4.0 * px * (x16.C + nout * Cout), ('conv', 'p24' if x16.fmt == K.FMT_BX3 else 'p16')))
            return out

        def timed_wg(x, dout, Cout, Cin, KH, **kw):
            if kw.get('per_image'):
                return orig_wg(x, dout, Cout, Cin, KH, **kw)
            e0, e1 = _ev()
            e0.record()
            out = orig_wg(x, dout, Cout, Cin, KH, **kw)
            e1.record()
            recs.append((2.0 * dout.shape[0] * Cout * Cin * KH * KH * dout.shape[2] * dout.shape[3], e0, e1,
                         4.0 * (dout.shape[0] * Cin * x.shape[2] * x.shape[3] + dout.numel()),
                         ('wgrad', KH if kw.get('stride', 1) == 1 else 's2')))
            return out

        def timed_wg16(x16, d16, **kw):
            e0, e1 = _ev()
            e0.record()
            out = orig_wg16(x16, d16, **kw)
            e1.record()
            px = x16.N * x16.H * x16.W
            recs.append((2.0 * px * d16.C * x16.C * 9, e0, e1, 4.0 * px * (x16.C + d16.C), ('wgrad', 'p24' if x16.fmt == K.FMT_BX3 else 'p16')))
            return out

        def timed_grp(reqs, seq=0, want_db=True):
            e0, e1 = _ev()
            e0.record()
            out = orig_grp(reqs, seq=seq, want_db=want_db)
            e1.record()
            fl = by = 0.0
            for (x, dout, Cout, Cin, _gate) in reqs:
                px = dout.shape[0] * dout.shape[2] * dout.shape[3]
                fl += 2.0 * px * Cout * Cin
                by += 4.0 * (x.numel() + dout.numel())
            recs.append((fl, e0, e1, by, ('wgrad', 'g1'), len(reqs)))
            return out
        # the fused NAFBlock chain kernels and the depthwise stencils (24 % + 6 % of the step): one record per launch, keyed by kernel and C
        chain_orig = {n: getattr(K, n) for n in ('naf_head_fwd', 'naf_tail_fwd', 'naf_tail_bwd', 'naf_head_bwd', 'dwsg_fwd', 'dwsg_bwd')}

        def _chain(name, work):
            fn = chain_orig[name]

            def wrapped(*args, **kw):
                e0, e1 = _ev()
                e0.record()
                out = fn(*args, **kw)
                e1.record()
                fl, by, Cc = work(*args, **kw)
                recs.append((fl, e0, e1, by, ('chain', name), Cc))
                return out
            return wrapped

        def _w_head_fwd(x, *r, **kw):
            n, c, h, w = x.shape
            return 2.0 * n * h * w * 2 * c * c, 4.0 * n * h * w * 4 * c, c

        def _w_tail_fwd(g, s_, x, *r, c_out=None, **kw):
            n, c, h, w = g.shape
            co = c if c_out is None else c_out
            return 2.0 * n * h * w * (3 * c * c + co * c), 4.0 * n * h * w * (6 * c + co), c

        def _w_tail_bwd(dout, gamma, t4, *r, w3tp=None, **kw):
            n, co, h, w = dout.shape
            c = t4.shape[1] // 2
            return (2.0 * n * h * w * (c * co + 2 * c * c + (c * c if w3tp is not None else 0)),
                    4.0 * n * h * w * (co + 6 * c + (c if w3tp is not None else 0)), c)

        def _w_head_bwd(dt1, x, *r, **kw):
            n, c, h, w = x.shape
            return 2.0 * n * h * w * 2 * c * c, 4.0 * n * h * w * 5 * c, c

        def _w_dw_fwd(t, *r, **kw):
            n, c2, h, w = t.shape
            return 0.0, 4.0 * n * h * w * (c2 + c2 // 2), c2 // 2

        def _w_dw_bwd(dg, t, *r, **kw):
            n, c2, h, w = t.shape
            return 0.0, 4.0 * n * h * w * (c2 // 2 + 2 * c2), c2 // 2
        for nm, wk in (('naf_head_fwd', _w_head_fwd), ('naf_tail_fwd', _w_tail_fwd), ('naf_tail_bwd', _w_tail_bwd), ('naf_head_bwd', _w_head_bwd),
                       ('dwsg_fwd', _w_dw_fwd), ('dwsg_bwd', _w_dw_bwd)):
            setattr(K, nm, _chain(nm, wk))
        K.conv_forward, K.conv3x3_p16, K.conv_wgrad, K.wgrad3x3_p16, K.wgrad1x1_group = timed, timed_p16, timed_wg, timed_wg16, timed_grp
        graph_was = getattr(model, 'use_hip_graph', False)
        model.use_hip_graph = False                       # instrumented step runs eagerly (a replayed graph makes no Python calls)
        from textualdegremoval_amd import engine as _E
        # one stream: an event pair around a launch times THAT launch, not a co-running pair -- the deferred leaves (incl. the grouped 1x1
        # weight gradients) run on the current stream in front of the MASA-encoder backward instead of beside it
        serial_was, _E.SERIAL_LEAVES = _E.SERIAL_LEAVES, True
        try:
            # The eager host loop issues launches more slowly than the GPU retires them; an event pair around a launch
            # would then also time the idle gap before it.  Park the GPU on a calibrated spin kernel so that the whole
            # step is queued before it starts executing: the event pairs then bracket back-to-back device work.
            # one un-timed eager step in the same configuration first: buffers this configuration allocates for the first time (the
            # workspace of the grouped weight gradients on the current stream) must not meet a parked GPU -- hipMalloc waits for it
            it += 1
            step(it)
            recs.clear()
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            torch.cuda._sleep(20_000_000)
            c1.record()
            torch.cuda.synchronize()
            per_cycle_ms = max(c0.elapsed_time(c1), 1e-3) / 20_000_000
            torch.cuda._sleep(int(min(1500.0, 12 * (dt / a.steps * 1e3)) / per_cycle_ms))
            it += 1
            step(it)
            # what an event pair costs by itself on this queue (two marker packets back to back, nothing between them): subtracted
            # from every pair below, so that avg_launch_ms is the kernel's duration as rocprofv3 --kernel-trace reports it
            empties = []
            for _ in range(32):
                e0, e1 = _ev()
                e0.record()
                e1.record()
                empties.append((e0, e1))
            torch.cuda.synchronize()
            ev_overhead_ms = sorted(e0.elapsed_time(e1) for e0, e1 in empties)[len(empties) // 2]
        finally:
            K.conv_forward, K.conv3x3_p16, K.conv_wgrad, K.wgrad3x3_p16, K.wgrad1x1_group = orig, orig_p16, orig_wg, orig_wg16, orig_grp
            for nm, fn in chain_orig.items():
                setattr(K, nm, fn)
            model.use_hip_graph = graph_was
            _E.SERIAL_LEAVES = serial_was
        # The 3x3 / stride-1 forward + data-gradient launches of the step are run by two kernels since round 4 (conv3x3_p16_kernel on
        # pre-split operands for C >= 64, the fp32-tensor kernel conv_bx3_kernel / conv_mfma_kernel for the C = 32 level): `roofline` is
        # the one with most device time, `roofline.family_3x3_s1` the all-launch aggregate (+ `by_kernel`), `roofline_other` the second
        # kernel and the weight-gradient families (kernel + its split-K reduction), all from the same instrumented step.
        PEAK_HX2 = PEAK_BF16 / 3.0
        CONV = {0: ('conv_mfma_kernel<KH=3,S=1> (exact fp32 v_mfma_f32_32x32x2_f32)', PEAK_F32, 'dense fp32 MFMA peak'),
                1: ('conv_bx3_kernel<KH=3,S=1,SCH_BX3> (3-way bf16 split, 6 x v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate)',
                    PEAK_BX3, '2.5 PFLOP/s dense bf16 MFMA / 6 cross products = fp32-equivalent peak of the split scheme'),
                2: ('conv_bx3_kernel<KH=3,S=1,SCH_HX2> (fp32 tensors in, 2-way fp16 split per consumer, 3 x v_mfma_f32_32x32x16_f16 per fp32 product)',
                    PEAK_HX2, '2.5 PFLOP/s dense f16 MFMA / 3 cross products = fp32-equivalent peak of the split scheme'),
                3: ('conv_bx3_kernel<KH=3,S=1,H1> (fp32 tensors in, operands rounded to ONE fp16 plane per consumer, 1 x v_mfma_f32_32x32x16_f16 per '
                    'product, fp32 accumulate -- the plain fp16-MFMA arithmetic of BASELINE configs[4])',
                    PEAK_BF16, '2.5 PFLOP/s dense f16 MFMA (one product per multiply)'),
                'p16': ('conv3x3_p16_kernel (pre-split fp16 pair planes in and out, both operands by LDS-DMA, 3 x v_mfma_f32_32x32x16_f16 per fp32 product)',
                        PEAK_HX2, '2.5 PFLOP/s dense f16 MFMA / 3 cross products = fp32-equivalent peak of the split scheme'),
                'p24': ('conv3x3_p16_kernel<PF_TRI> (pre-split bf16 TRIPLE planes in and out -- h + m + l == the fp32 value exactly --, both operands by '
                        'LDS-DMA, 6 x v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate)',
                        PEAK_BX3, '2.5 PFLOP/s dense bf16 MFMA / 6 cross products = fp32-equivalent peak of the split scheme')}

        def entry(rr, name, pk, note):
            fl = sum(r[0] for r in rr)
            ms = sum(max(r[1].elapsed_time(r[2]) - ev_overhead_ms, 1e-4) for r in rr)
            ach = fl / (ms * 1e-3) / 1e12
            by = sum(r[3] for r in rr)
            e = {'bound': 'mfma', 'kernel': name, 'achieved': ach, 'peak': pk / 1e12, 'unit': 'TFLOP/s', 'frac': ach / (pk / 1e12),
                 'peak_note': note, 'frac_of_f32_mfma_peak': ach / (PEAK_F32 / 1e12), 'launches': len(rr),
                 'avg_launch_ms': ms / max(len(rr), 1), 'alg_flop_per_launch': fl / max(len(rr), 1), 'traffic': None,
                 'alg_bytes_per_launch': by / max(len(rr), 1), '_total_ms': ms}
            if by > 0 and fl / by < pk / PEAK_HBM:
                # arithmetic intensity below the ridge of this ceiling (peak flop/s / 8 TB/s): the family is HBM-bound, price it on bytes
                gbs = by / (ms * 1e-3) / 1e9
                e.update({'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM / 1e9, 'unit': 'GB/s', 'frac': gbs / (PEAK_HBM / 1e9),
                          'peak_note': f'8 TB/s HBM3E; intensity {fl / by:.0f} flop/B < ridge {pk / PEAK_HBM:.0f} flop/B of the {pk / 1e12:.0f} TFLOP/s ceiling',
                          'achieved_tflops': ach, 'frac_of_split_flop_ceiling': ach / (pk / 1e12)})
            return e
        conv = [r for r in recs if r[4][0] == 'conv']
        subs = {k: entry([r for r in conv if r[4][1] == k], *CONV[k]) for k in sorted({r[4][1] for r in conv}, key=str)}
        if subs:
            # `roofline` is the DOMINANT KERNEL of the step (most device time): conv3x3_p16_kernel since round 4.  The other kernel(s) of
            # the 3x3 / stride-1 forward + data-gradient family lead `roofline_other`, and `family_3x3_s1` keeps the all-launch aggregate
            # rounds 1-3 reported (one kernel ran the whole family then), priced against the ceiling of the scheme that carries most of it.
            lead = max(subs, key=lambda k: subs[k]['_total_ms'])
            split = [k for k in subs if CONV[k][1] == CONV[lead][1]]          # kernels priced against the same ceiling
            prefixes = {0: 'conv_mfma_kernel<3, 1, 1,', 1: 'conv_bx3_kernel<3, 1,', 2: 'conv_bx3_kernel<3, 1,', 3: 'conv_bx3_kernel<3, 1,', 'p16': 'conv3x3_p16_kernel',
                        'p24': 'conv3x3_p16_kernel'}
            for k in subs:
                subs[k]['traffic'] = pmc_traffic([prefixes[k]])[0]
            roof = dict(subs[lead])
            roof['traffic_unit'] = pmc_traffic([prefixes[lead]])[1]
            roof['clock'] = pmc_clock(prefixes[lead].split('<')[0])
            fam = entry([r for r in conv if r[4][1] in split], ' + '.join(CONV[k][0] for k in split), CONV[lead][1], CONV[lead][2])
            fam['traffic'] = pmc_traffic(sorted({prefixes[k] for k in split}))[0]
            fam['by_kernel'] = [{kk: vv for kk, vv in subs[k].items() if kk != '_total_ms'} for k in split]
            fam.pop('_total_ms', None)
            roof['family_3x3_s1'] = fam
            roof['event_pair_overhead_ms'] = ev_overhead_ms
            roof_other = [subs[k] for k in subs if k != lead]
        WG = {3: 'wgrad_bx3_kernel<KH=3> + wgrad_reduce_kernel (fp32 tensors in, operand split + v_alignbit fragment assembly per consumer)',
              1: 'wgrad1x1_sp_kernel<KQ=1> (128 x 128 tiles: every operand value split once per workgroup, bf16 planes shared through LDS) / '
                 'wgrad1x1_dma_kernel<1,1> (64 x 64 tiles) + wgrad_reduce_kernel (1x1 weight gradients of the NAFBlock chains, split-K partials)',
              'p16': 'wgrad3x3_p16_kernel + wgrad_p16_reduce_kernel (pre-split pair planes, transposed LDS reads, no operand VALU)',
              'p24': 'wgrad3x3_p16_kernel<NS=3> + wgrad_p16_reduce_kernel (pre-split bf16 triple planes, transposed LDS reads, no operand VALU)',
              's2': 'wgrad_s2_kernel + wgrad_reduce_kernel (3x3 / 2x2 stride-2 level transitions: parity-de-interleaved LDS planes, 12 / 8-wave workgroups)',
              'g1': 'wgrad1x1_sp_kernel<GRP> + wgrad1x1_grp_reduce_kernel (tdr_wgrad1x1_group: the deferred 1x1 leaf weight gradients of one shape -- e.g. the '
                    '58 conv1 / conv4 problems of the C = 256 level -- in ONE launch + ONE fixed-order reduction; `launches` = grouped launches, '
                    '`problems` = weight gradients they carry)'}
        for k in sorted({r[4][1] for r in recs if r[4][0] == 'wgrad'}, key=str):
            e = entry([r for r in recs if r[4] == ('wgrad', k)], WG.get(k, f'wgrad KH={k}'),
                      PEAK_BF16 if K.MATH == 'h1' else (PEAK_HX2 if K.MATH == 'hx2' else (PEAK_BX3 if K.MATH == 'bx3' else PEAK_F32)),
                      'fp32-equivalent ceiling of the step\'s operand scheme; time = kernel + its fixed-order split-K reduction (HIP events around both)')
            # measured bytes per launch: the kernel's own traffic plus its split-K reduction's (one reduction per weight-gradient launch)
            wpre = {3: ('wgrad_bx3_kernel<3,', 'wgrad_reduce_kernel'), 1: ('wgrad1x1_', 'wgrad_reduce_kernel'),
                    'p16': ('wgrad3x3_p16_kernel', 'wgrad_p16_reduce_kernel'), 'p24': ('wgrad3x3_p16_kernel', 'wgrad_p16_reduce_kernel'),
                    's2': ('wgrad_s2_kernel', 'wgrad_reduce_kernel'), 'g1': ('wgrad1x1_sp_kernel<false, 0, 1, true', 'wgrad1x1_grp_reduce_kernel')}.get(k)
            if wpre:
                t_k, t_r = pmc_traffic([wpre[0]])[0], pmc_traffic([wpre[1]])[0]
                e['traffic'] = None if t_k is None else t_k + (t_r or 0.0)
            if k == 'g1':
                e['problems'] = sum(r[5] for r in recs if r[4] == ('wgrad', 'g1'))
            roof_other.append(e)
        CHAIN = {'naf_head_fwd': 'naf_head_fwd_kernel (norm1 -> conv1, one workgroup = 64 pixels x all channels)',
                 'naf_tail_fwd': 'naf_tail_fwd_kernel (conv3 -> +x -> norm2 -> conv4 -> SimpleGate -> conv5 -> +y)',
                 'naf_tail_bwd': 'naf_tail_bwd_kernel<HEAD=false> (conv5^T -> gate bwd -> conv4^T -> norm2 bwd -> conv3^T)',
                 'naf_head_bwd': 'naf_tail_bwd_kernel<HEAD=true> (conv1^T -> norm1 bwd + skip)',
                 'dwsg_fwd': 'dwsg_stencil_kernel (depthwise 3x3 + SimpleGate + SCA pool partials)',
                 'dwsg_bwd': 'dwsg_bwd_fused_kernel (one-pass depthwise + SimpleGate backward) + its parameter-gradient partials'}
        pk_split = PEAK_BF16 if K.MATH == 'h1' else (PEAK_HX2 if K.MATH == 'hx2' else (PEAK_BX3 if K.MATH == 'bx3' else PEAK_F32))
        for nm in CHAIN:
            rr = [r for r in recs if r[4] == ('chain', nm)]
            if not rr:
                continue
            e = entry(rr, CHAIN[nm], pk_split, 'fp32-equivalent ceiling of the step\'s operand scheme (chain GEMMs); bytes = the fp32 tensors the '
                      'launch reads and writes once (DESIGN 5)')
            e['by_channels'] = []
            for cc in sorted({r[5] for r in rr}):
                sub = entry([r for r in rr if r[5] == cc], CHAIN[nm], pk_split, '')
                e['by_channels'].append({'C': cc, 'launches': sub['launches'], 'avg_launch_ms': sub['avg_launch_ms'], 'bound': sub['bound'],
                                         'frac': sub['frac'], 'frac_of_split_flop_ceiling': sub.get('frac_of_split_flop_ceiling', sub['frac']),
                                         'hbm_frac': (sub['alg_bytes_per_launch'] / (sub['avg_launch_ms'] * 1e-3)) / PEAK_HBM})
            e['traffic'] = pmc_traffic([{'naf_head_fwd': 'naf_head_fwd_kernel', 'naf_tail_fwd': 'naf_tail_fwd_kernel', 'naf_tail_bwd': 'naf_tail_bwd_kernel<',
                                         'naf_head_bwd': 'naf_tail_bwd_kernel<', 'dwsg_fwd': 'dwsg_stencil_kernel', 'dwsg_bwd': 'dwsg_bwd_fused_kernel'}[nm]])[0] \
                if nm not in ('naf_tail_bwd', 'naf_head_bwd') else None
            roof_other.append(e)
        for f in ([roof] if roof else []) + roof_other:
            f.pop('_total_ms', None)

    if rank == 0:
        ips = world * a.batch * a.steps / dt
        per_gpu = ips / world
        is_cfg2 = (a.arch, a.width, enc, a.size, a.batch) == ('nafnet', 32, [1, 1, 1, 28], 512, 4)
        line = {
            'metric': f'train images/sec ({a.size}x{a.size}, bs={a.batch}/GPU)', 'value': ips, 'unit': 'images/sec', 'n_gpus': ranks_seen,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            # what the number is: tensors, accumulators, optimiser and reductions are fp32 in every mode; `dtype` names how the dense
            # contractions are evaluated on the matrix cores (the only place the modes differ)
            'dtype': {'hx2': 'f32 (2xfp16-split MFMA: 22-bit operand significands, fp32 accumulate; loss-scaled backward; the MASA-encoder '
                             'ResidualBlock activations / gradients are stored as the fp16 pair (head + residual, 4 bytes) instead of fp32)',
                      'bx3': 'f32 (3xbf16-split MFMA: 24-bit operand significands on the full fp32 exponent range, fp32 accumulate; unscaled '
                             'gradients, no loss scale, no step-skip guard; tensors fp32 -- the MASA-encoder ResidualBlock activations / gradients '
                             'as three bf16 planes whose sum IS the fp32 value, bit for bit)',
                      'h1': 'f16 (single fp16 MFMA product, fp32 accumulate; reduced precision)',
                      'f32': 'f32 (exact fp32 MFMA)'}[K.MATH],
            'data': 'synthetic',
            'guard': (None if g_after is None or g_after.growth_interval < 0 else
                      {'skipped_total': int(g_after.skipped),
                       'skipped_in_timed_region': int(g_after.skipped) - int(g_before.skipped if g_before is not None else 0),
                       'applied_steps': int(g_after.step),
                       'scale_log2': __import__('math').log2(g_after.scale) if g_after.scale > 0 else None,
                       'note': 'device-resident step guard of the loss-scaled fp16-split backward: a non-finite gradient norm skips the '
                               'optimiser step and halves the scale (the reference never skips); a timed region with skipped steps is '
                               'not a clean measurement'}),
            'arithmetic': {'bx3': 'the library default: fp32 tensors; dense contractions as 3-way bf16 split x = h + m + l (6 bf16 MFMA products per fp32 '
                            'product, fp32 accumulate): per-product error <= one fp32 rounding (profiles/r1/bf16x3_probe_mi355x.log), any fp32 exponent, '
                            'so the backward pass runs on the raw gradients (no loss scale) and every optimiser step is applied (no guard verdict), as '
                            'the reference (image_restoration_ref_model.py:268-279).  Where a tensor is only consumed by contractions (MASA-encoder '
                            'ResidualBlocks, C >= 64) the producer stores the three planes instead of the fp32 value -- exactly the same number '
                            '(csrc/tdr_conv_p16.hip PF_TRI; tests/test_hip_p24.py: bit-identical to the fp32-tensor kernels).  TDR_MATH=f32 selects '
                            'exact fp32 MFMA, TDR_MATH=hx2 the disclosed fast mode',
                     'hx2': 'fp32 tensors; dense contractions as 2-way fp16 split (3 f16 MFMA products per fp32 product, fp32 accumulate), '
                            'the backward pass on gradients scaled by an exact power of two (dpred ~ 2^9, removed when the parameter '
                            'gradients are gathered); the 3x3 ResidualBlock convolutions of the MASA encoder (C >= 64) read and write their '
                            'activations / gradients pre-split (csrc/tdr_conv_p16.hip: the pair IS the stored tensor there, 22-23 significant '
                            'bits incl. the residual stream; engine.P16_ON = False restores fp32 tensors + per-consumer split); '
                            'per-image correlations as 3-way bf16 split / exact fp32: measured error of the split '
                            'schemes = that of the exact fp32 MFMA chain (profiles/r1/fp16x2_probe_mi355x.log, bf16x3_probe_mi355x.log, '
                            'grad_range_survey_cfg2.log); TDR_MATH=bx3 / f32 select the all-bf16-split / exact fp32 MFMA paths',
                     'h1': 'fp32 tensors; dense contractions on plain fp16 MFMA (operands rounded to one fp16 plane, ONE product, fp32 '
                           'accumulate) with the loss-scaled backward pass -- REDUCED precision, the "fp16 MFMA" arithmetic of BASELINE '
                           'configs[4]; not the headline arithmetic',
                     'f32': 'exact fp32 MFMA (v_mfma_f32_32x32x2_f32)'}[K.MATH],
            'config': {'workload': ((('BASELINE configs[1]: ' if is_cfg2 else '') +
                                     f"NAFNet-width{a.width} enc{str(enc).replace(' ', '')} + ref fusion [2,2,2,2,2], " +
                                     f'{a.size}x{a.size} color denoise sigma=15, bs={a.batch}/GPU, fwd+L1+bwd+clip+AdamW')
                                    if a.arch == 'nafnet' else
                                    (('BASELINE configs[2] per-GPU workload: ' if (a.size, a.batch) == (256, 8) else
                                      'BASELINE configs[4] per-GPU shape (512x512, bs 2): ' if (a.size, a.batch) == (512, 2) else '') +
                                     'Restormer-ref dim48 blocks[4,6,6,8] refine4 heads[1,2,4,8] '
                                     f'fusion[2,2,2,2], {a.size}x{a.size} synthetic pairs, bs={a.batch}/GPU, fwd+L1+bwd+clip+AdamW'
                                     if a.arch == 'restormer' else
                                     'DRSformer-ref without MEFC (007_drsformer_image_deraining_rain200l.yml network): dim48 blocks[4,6,6,8] '
                                     f'top-k sparse attention + mixed-scale FFN, {a.size}x{a.size} synthetic pairs, bs={a.batch}/GPU, fwd+L1+bwd+clip+AdamW'
                                     if a.arch == 'drsformer' else
                                     'DRSformer-ref with MEFC (008/009/010_drsformer_*.yml network): dim48 blocks[4,6,6,8], '
                                     f'{a.size}x{a.size} synthetic pairs, bs={a.batch}/GPU, fwd+L1+bwd+clip+AdamW'
                                     if a.arch == 'drsformer_mefc' else
                                     'PromptIR-ref (001_promptir_all_in_one_restoration.yml network, decoder=True): dim48 blocks[4,6,6,8] '
                                     f'refine4 prompts 64/128/320, {a.size}x{a.size} synthetic pairs, bs={a.batch}/GPU, fwd+L1+bwd+clip+AdamW')),
                       'width': a.width if a.arch == 'nafnet' else 48, 'enc_blk_nums': enc if a.arch == 'nafnet' else [4, 6, 6, 8],
                       'global_batch': world * a.batch,
                       'parallelism': f'dp{world}',
                       'collectives': ('none (1 GPU)' if world == 1 else
                                       ('tdr_comm_* (RCCL through the C ABI)' if comm is not None else f'torch.distributed ({a.backend})')),
                       'ranks_seen': ranks_seen,
                       'streams': ('2: the leaf 1x1 weight gradients of the blocks run deferred on a second HIP stream beside the MASA-encoder '
                                   'backward (engine.DEFER_WGRAD; the roofline leg times its launches on one stream)'
                                   if a.arch in ('nafnet', 'restormer', 'promptir', 'drsformer', 'drsformer_mefc') and world == 1 and os.environ.get('TDR_FORCE_DP_SCHEDULE', '0') != '1' else '1'),
                       'grad_exchange': ('none' if world == 1 else
                                         (f'{len(red.buckets)} buckets of <= 64 MiB, each all-reduced on the comm stream between the segments of '
                                          f'the captured backward ({red.bucket_launches} bucket exchanges issued so far)'
                                          if getattr(model, '_gstate', None) and model._gstate.get('split') else
                                          'one flat all-reduce between the captured graphs' if getattr(model, 'use_hip_graph', False)
                                          else 'per-bucket all-reduce overlapped with the eager backward')),
                       'dino_match': ('skipped bit-identically (ref size == lq size, N=1 window)' if a.dino_ref_size <= a.size else
                                      f'DINOv2 ViT-B/14 window match every step, ref {a.dino_ref_size}x{a.dino_ref_size} '
                                      f'({((a.dino_ref_size - a.size) // max(a.size // 4, 1) + 1) ** 2} windows/image, random-init ViT)')},
            'final_loss': loss,
            'hbm_peak_allocated_gb': torch.cuda.max_memory_allocated() / 1e9,     # of 288 GB (saved activations + deferred gradient operands + arena)
        }
        if is_cfg2:
            nprod = {'hx2': 3.0, 'bx3': 6.0, 'h1': 1.0}.get(K.MATH)
            line['roofline_step'] = {'achieved_hbm_frac': CFG2['B_alg'] * per_gpu / PEAK_HBM,
                                     'achieved_f32_flop_frac': CFG2['F_alg'] * per_gpu / PEAK_F32,
                                     'alg_bytes_per_image': CFG2['B_alg'], 'alg_flop_per_image': CFG2['F_alg']}
            if nprod:      # fp32-equivalent ceiling of the split scheme in use: 2.5 PFLOP/s dense f16 / bf16 MFMA over its products
                line['roofline_step']['achieved_split_flop_frac'] = CFG2['F_alg'] * per_gpu / (PEAK_BF16 / nprod)
                line['roofline_step']['split_peak_tflops'] = PEAK_BF16 / nprod / 1e12
            # what the counters say really moves per step (all kernels, PMC passes of profiles/pmc_collect.sh) against the same clock
            mb, mpath = pmc_step_bytes()
            if mb:
                line['roofline_step']['measured_hbm_bytes'] = mb
                line['roofline_step']['measured_hbm_frac'] = mb / (dt / a.steps) / PEAK_HBM
                line['roofline_step']['measured_hbm_source'] = mpath + ' (FETCH_SIZE + WRITE_SIZE over every kernel of an eager step)'
        if scale_diag is not None:
            line['scale_diagnostics'] = scale_diag
        if roof is not None:
            line['roofline'] = roof
            if roof_other:
                line['roofline_other'] = roof_other
        if a.arch == 'restormer':
            px = (a.size / 256.0) ** 2                     # SURVEY 8(d) quotes the figures per 256x256 image
            CFG3 = dict(B_alg=69.4e9 * px, F_alg=1.86e12 * px)
            line['roofline_step'] = {'achieved_hbm_frac': CFG3['B_alg'] * per_gpu / PEAK_HBM,
                                     'achieved_f32_flop_frac': CFG3['F_alg'] * per_gpu / PEAK_F32,
                                     'alg_bytes_per_image': CFG3['B_alg'], 'alg_flop_per_image': CFG3['F_alg']}
            if K.MATH == 'h1':
                # SURVEY 8(d) quotes cfg5's bytes on the fp32 basis "(/ 2 for fp16 activations)".  This library's h1 mode rounds the OPERANDS of
                # the contractions to fp16 inside the kernels; the tensors in HBM stay fp32 (the backward pass, LayerNorm statistics and the
                # optimiser read them), so the fp32 basis is the traffic this step really has -- the halved basis is given beside it
                line['roofline_step']['achieved_hbm_frac_fp16_activation_basis'] = 0.5 * CFG3['B_alg'] * per_gpu / PEAK_HBM
                line['roofline_step']['note'] = ('h1 = plain fp16 MFMA operands, fp32 accumulate, fp32 tensors in HBM: achieved_hbm_frac is on the '
                                                 'fp32 byte basis the step actually moves; SURVEY 8d\'s halved (fp16-activation) basis beside it')
        if not a.no_f32_exact and world == 1 and K.MATH != 'f32':
            line['f32_exact'] = f32_exact_run(a)
            if K.MATH == 'bx3' and is_cfg2:
                line['fast_mode'] = fast_mode_run(a)
        if not a.no_matcher_active and world == 1 and is_cfg2 and a.dino_ref_size <= a.size:
            line['matcher_active'] = matcher_active_run(a)
        if not a.no_cpu_baseline and world == 1 and a.arch == 'nafnet':
            line['cpu_baseline'] = cpu_baseline(a.width, enc, a.size, a.batch)
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
