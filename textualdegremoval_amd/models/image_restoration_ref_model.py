"""RefGuidedImageCleanModel on the HIP engine -- the train/step API of the
reference's models/image_restoration_ref_model.py (:56-438) with the same
method names, config keys and error behaviour, so
scripts/train/main_train_restoration_with_ref_input.py drives it unchanged.

optimize_parameters = DINO window match (bit-identically skipped when ref and lq
have equal size: N==1 window, top-1 of one) -> forward -> L1 -> hand-written
backward -> RCCL gradient average (dist) -> global-norm clip(0.01) + AdamW,
all on libtdr_hip.so kernels."""
import importlib
import math
import os
from collections import OrderedDict
from copy import deepcopy
from functools import partial

import torch

from .. import engine as E
from .. import kernels as K
from .. import losses as loss_module
from .. import metrics as metric_module
from ..metrics import tensor2img
from ..optim import FusedClipAdamW
from .archs import define_network
from .base_model import BaseModel, logger


GRAD_SCALE = True       # module switch: the exact power-of-two loss scale of the fp16 arithmetics (hx2 / h1)

class RefGuidedImageCleanModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.net_g = define_network(deepcopy(opt['network_g']))
        self.net_g = self.model_to_device(self.net_g)
        self.print_network(self.net_g)
        # DINOv2 ViT-B/14 window matcher (reference :74-90: vit_base(518, 14, init_values=1.0, 'mlp'), strict-loaded
        # from path.pretrain_dino, frozen).  Only exercised when ref is larger than lq.
        self.net_ext = None
        self.pretrain_dino = self.opt['path'].get('pretrain_dino')
        if self.pretrain_dino is not None and self.device.type == 'cuda':
            from ..dino import DinoMatcher
            self.net_ext = DinoMatcher(torch.load(self.pretrain_dino, map_location='cpu'), self.device, patch=14, heads=12)
        load_path = self.opt['path'].get('pretrain_network_g', None)
        if load_path is not None:
            self.load_network(self.net_g, load_path, self.opt['path'].get('strict_load_g', False),
                              param_key=self.opt['path'].get('param_key', 'params'))
        if self.is_train:
            self.init_training_settings()

    def init_training_settings(self):
        self.net_g.train()
        train_opt = self.opt['train']
        self.ema_decay = train_opt.get('ema_decay', 0)
        if self.ema_decay > 0:
            self.net_g_ema = define_network(deepcopy(self.opt['network_g'])).to(self.device)
            load_path = self.opt['path'].get('pretrain_network_g', None)
            if load_path is not None:
                self.load_network(self.net_g_ema, load_path, self.opt['path'].get('strict_load_g', True), 'params_ema')
            else:
                self.model_ema(0)
            self.net_g_ema.eval()
        if train_opt.get('pixel_opt'):
            pixel_type = train_opt['pixel_opt'].pop('type')
            self.cri_pix = getattr(loss_module, pixel_type)(**train_opt['pixel_opt']).to(self.device)
        else:
            raise ValueError('pixel loss are None.')
        self.setup_optimizers()
        self.setup_schedulers()

    def _optimizer_named_parameters(self):
        """ALL named_parameters(), in registration order, like the reference's setup_optimizers (:160-170).  Tensors an
        architecture registers but never uses (PromptIR-ref's chnl_reduce* / reduce_noise_channel_*, DRSformer-200L's
        masa_blk_enc_level1.*) stay in the param groups exactly as there: their .grad stays None, so clip_grad_norm_ and
        AdamW skip them (FusedClipAdamW._build does the same) -- and the parameter INDICES of optimizer.state_dict() are the
        reference's, so its `.state` resume files load aligned."""
        return list(self.net_g.named_parameters())

    def setup_optimizers(self):
        train_opt = self.opt['train']
        # reference quirk R3: the key read is 'fix_iterations' (YAMLs set 'param_fix_iterations')
        self.param_fix_iters = train_opt['fix_iterations'] if 'fix_iterations' in train_opt else None
        optim_params, optim_ref_params = [], []
        for k, v in self._optimizer_named_parameters():
            (optim_ref_params if 'masa' in k else optim_params).append(v)
        groups = [{'params': optim_params, 'lr': train_opt['optim_g']['lr']},
                  {'params': optim_ref_params, 'lr': train_opt['optim_g']['ref_lr']}]
        optim_type = train_opt['optim_g'].pop('type')
        clip = bool(train_opt.get('use_grad_clip', False))
        if optim_type == 'AdamW':
            if self.device.type == 'cuda':
                self.optimizer_g = FusedClipAdamW(groups, lr=train_opt['optim_g']['lr'],
                                                  weight_decay=train_opt['optim_g']['weight_decay'],
                                                  betas=train_opt['optim_g']['betas'], max_norm=0.01, use_grad_clip=clip)
            else:       # host-only objects (scheduler tables, checkpoint plumbing); optimize_parameters refuses to run
                self.optimizer_g = torch.optim.AdamW(groups, lr=train_opt['optim_g']['lr'],
                                                     weight_decay=train_opt['optim_g']['weight_decay'],
                                                     betas=train_opt['optim_g']['betas'])
        elif optim_type == 'Adam':
            # reference :176-178 `torch.optim.Adam(groups, **train_opt['optim_g'])` (its YAMLs never select it): the same
            # multi-tensor kernel with the L2 term in the gradient; keys other than lr/betas/eps/weight_decay are rejected
            kw = {k: v for k, v in train_opt['optim_g'].items() if k != 'ref_lr'}
            extra = set(kw) - {'lr', 'betas', 'eps', 'weight_decay'}
            if extra:
                raise NotImplementedError(f'Adam options {sorted(extra)} are not supported on the HIP optimiser')
            if self.device.type == 'cuda':
                self.optimizer_g = FusedClipAdamW(groups, lr=kw['lr'], weight_decay=kw.get('weight_decay', 0.0),
                                                  betas=kw.get('betas', (0.9, 0.999)), eps=kw.get('eps', 1e-8), max_norm=0.01,
                                                  use_grad_clip=clip, coupled_decay=True)
            else:
                self.optimizer_g = torch.optim.Adam(groups, **kw)
        else:
            raise NotImplementedError(f'optimizer {optim_type} is not supperted yet.')
        self.optimizers.append(self.optimizer_g)

    def feed_train_data(self, data):
        self.lq = data['lq'].to(self.device)
        if 'gt' in data:
            self.gt = data['gt'].to(self.device)
        if 'ref' in data:
            self.ref = data['ref'].to(self.device)

    feed_data = feed_train_data

    # ------------------------------------------------------------------ step
    def _match_reference_window(self):
        """reference :215-247.  With ref.shape == lq.shape the unfold yields one window and top-1 of one candidate is
        that window: ref_in == ref bit-exactly, so the two frozen ViT passes (whose only output is that arg-max) are
        skipped.  Otherwise the DINOv2 matcher picks the most similar lq-sized window of ref (HIP kernels, no-grad)."""
        if self.ref.shape[-2:] == self.lq.shape[-2:]:
            return self.ref
        if self.net_ext is None:
            raise ValueError('ref is larger than lq: the DINOv2 window matcher needs path.pretrain_dino '
                             '(the reference loads it unconditionally, image_restoration_ref_model.py:83-86)')
        ref_in, self.match_index, self.match_corr = self.net_ext.match(self.lq, self.ref)
        return ref_in

    def optimize_parameters(self, current_iter):
        kind = self.cri_pix.step_kind() if hasattr(self.cri_pix, 'step_kind') else None
        fused = self.device.type == 'cuda' and kind is not None and isinstance(self.optimizer_g, FusedClipAdamW)
        if not fused:
            raise NotImplementedError('HIP step needs a GPU, a mean-reduced pixel criterion of losses/ and the fused optimiser')
        # reference :205-212: while current_iter < fix_iterations the "masa" parameters get requires_grad_(False) -- and,
        # as written there, nothing ever turns them back on (the `else` belongs to `param_fix_iters is not None`).  A
        # frozen tensor has grad None: clip_grad_norm_ and AdamW (decay included) skip it.  Here: param group 1 frozen.
        if self.param_fix_iters is not None and current_iter < self.param_fix_iters:
            self._masa_frozen = True
        self.optimizer_g.set_frozen_groups({1} if getattr(self, '_masa_frozen', False) else set())
        self.ref_in = self._match_reference_window()
        if not hasattr(self, '_step_names'):
            net = self.get_bare_model(self.net_g)
            unused = tuple(getattr(net, 'unused_parameter_prefixes', ()))
            named = [(k, p) for k, p in net.named_parameters() if not (unused and k.startswith(unused))]
            self._step_names = [k for k, _ in named]
            self._step_params = [p for _, p in named]
            if not hasattr(self, 'grad_reducer'):
                from ..parallel import GradAllReducer
                self.grad_reducer = GradAllReducer(list(zip(self._step_names, self._step_params)))
            self.use_hip_graph = os.environ.get('TDR_GRAPH', '1') == '1'
            self._gstate = None
        self.optimizer_g.use_grad_clip = bool(self.opt['train']['use_grad_clip'])
        K.set_group_owner(id(self))            # pinned pointer tables of the grouped leaf weight gradients: one ring per model
        if self._survey_due(current_iter):
            loss = self._surveyed_step(current_iter)
        elif self.use_hip_graph:
            loss = self._graph_step()
        else:
            loss = self._eager_step(self.lq, self.gt, self.ref_in)
        self.log_dict = self.reduce_loss_dict(OrderedDict(l_pix=loss[0]))
        if self.ema_decay > 0:
            self.model_ema(decay=self.ema_decay)

    # ---- fp16-window survey (TDR_MATH=hx2).  The 2-way fp16 split needs its operands inside +-65504 with their leading
    # magnitudes well above 2^-14; forward activations of these networks sit there (LayerNorm-ed, O(1)), the backward
    # pass is put there by the loss scale.  Every TDR_RANGE_CHECK_EVERY steps (and on the first step of a shape) the step
    # runs eagerly with kernels.RangeSurvey installed: max|x| of every split operand, one host read after the step.
    #   gradients too large / too small for the window -> the loss scale moves by the measured number of binades
    #   gradient operands span more binades than the window holds, or an activation exceeds 2^15 -> the affected pass
    #   leaves the fp16 split for the 3-way bf16 split (full fp32 range) -- logged, graphs re-captured.
    # Between surveys the step guard (skip + halve on a non-finite norm) catches what drifts out at the top.
    GRAD_WINDOW = (-10, 14)          # allowed floor(log2 max|g|) of the scaled gradient operands
    FWD_MAX_EXP = 14

    def _survey_due(self, current_iter):
        if not K.fp16_path() or int(os.environ.get('TDR_RANGE_CHECK_EVERY', '1000')) <= 0:
            return False
        every = int(os.environ.get('TDR_RANGE_CHECK_EVERY', '1000'))
        last = getattr(self, '_last_survey_iter', None)
        return last is None or current_iter - last >= every

    def _surveyed_step(self, current_iter):
        self._last_survey_iter = current_iter
        survey = K.RangeSurvey(self.lq.device)
        with survey:
            loss = self._eager_step(self.lq, self.gt, self.ref_in)
        r = survey.read()
        self.last_range_survey = r
        fmin, fmax, fbad = r['fwd']
        gmin, gmax, gbad = r['grad']
        lo, hi = self.GRAD_WINDOW
        changed = False
        if fbad or (fmax is not None and fmax > self.FWD_MAX_EXP):
            logger.warning(f'fp16-window survey (iter {current_iter}): forward operand with max|x| >= 2^{fmax} '
                           f'({fbad} non-finite): leaving the fp16 split, TDR_MATH=bx3 from here on')
            K.set_math('bx3')
            changed = True
        elif gmax is not None and not getattr(self, '_bwd_full_range', False):
            guard = self.optimizer_g.guard.read()
            shift = 0
            if gmax > hi:
                shift = hi - gmax
            elif gmin < lo:
                shift = min(lo - gmin, hi - gmax)
            if gmax - gmin > hi - lo or gbad:
                logger.warning(f'fp16-window survey (iter {current_iter}): scaled gradient operands span 2^{gmin}..2^{gmax} '
                               f'({gbad} non-finite): the backward pass leaves the fp16 split (3-way bf16 split, unscaled)')
                self._bwd_full_range = True
                changed = True
            elif shift != 0:
                new = guard.scale * 2.0 ** shift
                logger.warning(f'fp16-window survey (iter {current_iter}): scaled gradient operands span 2^{gmin}..2^{gmax}; '
                               f'loss scale 2^{math.log2(guard.scale):.0f} -> 2^{math.log2(new):.0f}')
                self._scale_shift = getattr(self, '_scale_shift', 0) + shift
                self.optimizer_g.guard.write(scale=new, max_scale=new)
                self._last_survey_iter = None         # starved operands flush to zero and hide the tensors behind them:
                #                                       look again on the next step until the window holds
        if changed:
            self._last_survey_iter = None
            self._gstate = None                       # graphs captured under the old arithmetic are stale
            self._pack_plan = K.PackPlan()
        return loss

    def _fwd_bwd(self, lq, gt, ref_in, defer_collectives=False, on_bucket=None):
        """forward, L1, hand-written backward; gradients land in the reducer's arena (RCCL-averaged when
        distributed, unless `defer_collectives`; `on_bucket`: GradAllReducer.begin).  Returns the loss tensor [1]."""
        net = self.get_bare_model(self.net_g)
        P = {k: p.data for k, p in zip(self._step_names, self._step_params)}
        if not hasattr(self, '_pack_plan'):
            self._pack_plan = K.PackPlan()
        prev_plan = K.set_pack_plan(self._pack_plan)
        prev_scaled = K.GRAD_SCALED
        try:
            kind, lw, eps = self.cri_pix.step_kind()
            # exact (power-of-two) loss scale for the fp16-split data-gradient kernels: dpred = S*lw/numel ~ 2^9, which puts
            # max|g| of every gradient operand of the step between ~2^-1 and 2^10 (profiles/grad_range_survey.py)
            # The scale lives in the optimiser's device-resident StepGuard: a non-finite gradient norm (an operand left the
            # fp16 range) skips that step and halves it, 1000 finite steps double it again up to this starting value.
            gs = 1.0
            if K.fp16_path() and GRAD_SCALE and lw > 0 and \
                    not getattr(self, '_bwd_full_range', False):
                # PSNRLoss: dpred ~ lw*(10/ln10)*2d / (N*CHW*mse_n), ~2^8 above an L1 gradient at d ~ 0.1: start lower
                gs = 2.0 ** (math.floor(math.log2(512.0 * lq.shape[0] * 3 * lq.shape[2] * lq.shape[3] / lw)) +
                             getattr(self, '_scale_shift', 0) - (8 if kind in (K.LOSS_PSNR, K.LOSS_PSNR_Y) else 0))
            K.set_grad_scaled(gs != 1.0)
            guard = self.optimizer_g.ensure_guard(lq.device)
            if not torch.cuda.is_current_stream_capturing():
                # TDR_MATH=bx3 / f32 (fp32 range in BOTH passes): no skip verdict -- the step is applied whatever the norm, as the
                # reference does (:276-279); the struct only counts steps on the device.  Under an fp16 arithmetic the verdict stays
                # even without a loss scale (the surveyed full-range backward, GRAD_SCALE = False): the FORWARD pass still runs inside
                # the fp16 window, and between surveys the guard is what keeps a forward overflow out of the weights
                guard.set_never_skip(gs == 1.0 and not K.fp16_path())
                guard.set_max_scale(gs)        # (host read; the capture pass reuses the value of the eager warm-up steps)
            self.grad_reducer.guard = guard
            self._pack_plan.run()              # all weights, all layouts, one launch (no-op on the recording step)
            eng = getattr(net, 'engine', E)    # RestormerRefFusion carries restormer_engine
            out, saved = eng.net_fwd(P, net.cfg, lq, ref_in)
            self.output = out
            loss, dpred = K.pixel_loss(kind, out.contiguous(), gt.contiguous(), lw, eps, guard=guard)
            sink = self.grad_reducer.begin(defer_collectives=defer_collectives, on_bucket=on_bucket)
            K.BACKWARD_PHASE = True
            try:
                eng.net_bwd(dpred, P, net.cfg, saved, G=sink)
            finally:
                K.BACKWARD_PHASE = False
            grads = self.grad_reducer.finish()
        finally:
            K.set_grad_scaled(prev_scaled)
            K.set_pack_plan(prev_plan)
            self._pack_plan.invalidate()       # the optimiser is about to change the weights
        if not getattr(self, '_grads_bound', False) or self.grad_reducer.relaid:
            for k, p in zip(self._step_names, self._step_params):
                p.grad = grads[k]
            self._grads_bound = True
        return loss

    @staticmethod
    def _raise_on_uncovered(red, missing, where):
        """An incomplete bucket was never GATHERED either: its arena slice still holds whatever an earlier step left there, and a
        flat all-reduce of the arena would hand those stale numbers to AdamW on every replay (consistent across ranks, and wrong).
        No silent fallback: a trainable parameter that receives no gradient is a configuration error (the reference's
        `find_unused_parameters`, base_model.py:77-82, is served by `unused_parameter_prefixes`: those parameters are frozen)."""
        if not missing:
            return
        names = [n for bi in missing for n in red.buckets[bi][2] if n not in getattr(red, '_pending', {})]
        raise RuntimeError(f'{where}: gradient bucket(s) {missing} were not completed by the backward pass -- '
                           f'no gradient arrived for {names[:8]}{" ..." if len(names) > 8 else ""}; freeze those parameters '
                           '(requires_grad=False / unused_parameter_prefixes) or run with TDR_GRAPH_BUCKETS=0 / TDR_GRAPH=0')

    def _eager_step(self, lq, gt, ref_in):
        loss = self._fwd_bwd(lq, gt, ref_in)
        self.optimizer_g.step()
        return loss

    def _graph_step(self):
        """The step as captured hipGraphs (forward+backward | clip+AdamW): ~1500 kernel launches are replayed without
        host work.  Shapes are static per graph; the first two steps of a shape run eagerly (allocator / workspace /
        arena-layout warm-up).
        Distributed: DDP overlaps bucketed all-reduces with the backward pass (reference models/base_model.py:76-82), and
        so does the captured step -- the forward+backward capture is CUT where a gradient bucket has just been gathered
        into the arena (GradAllReducer `on_bucket`), giving segments A_0 .. A_k that share one memory pool; the replay
        enqueues A_0, the RCCL all-reduce of bucket 0 on the comm stream (behind an event of the compute stream), A_1,
        bucket 1, ... so every exchange but the last runs under the remaining backward segments; graph B (clip + AdamW)
        waits for the comm stream.  TDR_GRAPH_BUCKETS=0 restores the single graph + one flat all-reduce."""
        key = (tuple(self.lq.shape), tuple(self.gt.shape), tuple(self.ref_in.shape), self.optimizer_g.use_grad_clip,
               tuple(sorted(self.optimizer_g.frozen_groups)))
        red = self.grad_reducer
        st = self._gstate
        if st is None or st['key'] != key:
            st = self._gstate = {'key': key, 'eager_left': 2, 'segs': None}
        if st['segs'] is None:
            if st['eager_left'] > 0:
                st['eager_left'] -= 1
                loss = self._eager_step(self.lq, self.gt, self.ref_in)
                if red.collective and os.environ.get('TDR_GRAPH_BUCKETS', '1') == '1':
                    # the same check the capture pass makes, on the FIRST eager step of the shape: a trainable parameter without a
                    # gradient is a configuration error of the run and should not surface two steps (and one capture) later
                    self._raise_on_uncovered(red, red.uncovered_buckets(), 'eager warm-up step')
                return loss
            st['lq'], st['gt'], st['ref'] = self.lq.clone(), self.gt.clone(), self.ref_in.clone()
            # what torch.cuda.graph() does on entry: hand the eager steps' cached blocks back, or the graphs' private pool has to
            # fit NEXT to them (PromptIR-ref 384x384 bs 8: 106 GB live + 180 GB cached = out of memory)
            torch.cuda.synchronize()
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            st['ws_refs'] = []                 # scratch buffers the captured kernels address (kernels.workspace_capture)
            split = red.collective and os.environ.get('TDR_GRAPH_BUCKETS', '1') == '1'
            pool = torch.cuda.graph_pool_handle()
            segs = []                          # [(graph, bucket index to exchange after it | None)]
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            # thread_local: the RCCL watchdog thread may touch the HIP runtime while this thread captures
            with torch.cuda.stream(cap), K.workspace_capture(st['ws_refs']):
                cur = [torch.cuda.CUDAGraph()]
                cur[0].capture_begin(pool=pool, capture_error_mode='thread_local')

                tail = [None]

                def cut(bi):
                    if not any(red._bucket_left):          # the last bucket: nothing follows it, no (empty) segment after it
                        tail[0] = bi
                        return
                    cur[0].capture_end()
                    segs.append((cur[0], bi))
                    cur[0] = torch.cuda.CUDAGraph()
                    cur[0].capture_begin(pool=pool, capture_error_mode='thread_local')
                try:
                    st['loss'] = self._fwd_bwd(st['lq'], st['gt'], st['ref'], defer_collectives=True,
                                               on_bucket=cut if split else None)
                finally:
                    cur[0].capture_end()
                segs.append((cur[0], tail[0]))
                st['pinned'] = red.pinned_tables      # host blocks the captured table uploads re-read on replay
                if split:
                    # every bucket must have been cut into the segment list: one whose parameters did not all receive a gradient in
                    # the captured step never completes, is never exchanged on replay, and the replicas would silently diverge
                    seen = {bi for _, bi in segs if bi is not None}
                    missing = red.uncovered_buckets() or sorted(set(range(len(red.buckets))) - seen)
                    self._raise_on_uncovered(red, missing, 'captured step')
                self.optimizer_g.prepare()
                gB = torch.cuda.CUDAGraph()
                gB.capture_begin(pool=pool, capture_error_mode='thread_local')
                try:
                    self.optimizer_g.launch()
                finally:
                    gB.capture_end()
            torch.cuda.current_stream().wait_stream(cap)
            st['segs'], st['gB'], st['split'] = segs, gB, split
            st['output'] = self.output
        else:
            st['lq'].copy_(self.lq, non_blocking=True)
            st['gt'].copy_(self.gt, non_blocking=True)
            st['ref'].copy_(self.ref_in, non_blocking=True)
            self.optimizer_g.prepare()
        for g, bi in st['segs']:
            g.replay()
            if bi is not None:
                red.launch_bucket(bi)
        if st['split']:
            red.wait_buckets()
        else:
            red.allreduce_flat()
        st['gB'].replay()
        self.output = st['output']
        return st['loss']

    skipped_steps = 0

    def get_current_log(self):
        """the base class's floats (a non-finite loss raises there); also the place where skipped optimiser steps become
        visible to the host: the step guard lives in device memory and is only read here, at print_freq."""
        out = super().get_current_log()
        opt = getattr(self, 'optimizer_g', None)
        if isinstance(opt, FusedClipAdamW) and opt.guard is not None:
            g = opt.guard.read()
            if g.skipped != self.skipped_steps:
                logger.warning(f'{g.skipped - self.skipped_steps} optimiser step(s) skipped since the last log: non-finite '
                               f'gradient norm (fp16-split backward pass out of range); loss scale now 2^{math.log2(g.scale):.0f}, '
                               f'{g.step} steps applied.  TDR_MATH=bx3 runs the backward pass with the full fp32 range.')
            self.skipped_steps = g.skipped
        return out

    # ------------------------------------------------------------------ validation (reference :286-409)
    def pad_test(self, window_size):
        import torch.nn.functional as F
        scale = self.opt.get('scale', 1)
        _, _, h, w = self.lq.size()
        mod_pad_h = (window_size - h % window_size) % window_size
        mod_pad_w = (window_size - w % window_size) % window_size
        img = F.pad(self.lq, (0, mod_pad_w, 0, mod_pad_h), 'reflect')
        self.nonpad_test(img)
        _, _, h, w = self.output.size()
        self.output = self.output[:, :, 0:h - mod_pad_h * scale, 0:w - mod_pad_w * scale]

    def nonpad_test(self, img=None):
        img = self.lq if img is None else img
        net = self.net_g_ema if hasattr(self, 'net_g_ema') else self.net_g
        net.eval()
        with torch.no_grad():
            pred = net(img, self.ref)
        self.output = pred[-1] if isinstance(pred, list) else pred
        if net is self.net_g:
            self.net_g.train()

    def dist_validation(self, dataloader, current_iter, tb_logger, save_img, rgb2bgr, use_image):
        if os.environ.get('LOCAL_RANK', '0') == '0':
            return self.nondist_validation(dataloader, current_iter, tb_logger, save_img, rgb2bgr, use_image)
        return 0.

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img, rgb2bgr, use_image):
        with_metrics = self.opt['val'].get('metrics') is not None
        if with_metrics:
            self.metric_results = {m: 0 for m in self.opt['val']['metrics'].keys()}
        window_size = self.opt['val'].get('window_size', 0)
        test = partial(self.pad_test, window_size) if window_size else self.nonpad_test
        cnt = 0
        for val_data in dataloader:
            self.feed_data(val_data)
            test()
            visuals = self.get_current_visuals()
            sr_img = tensor2img(visuals['result'], rgb2bgr=rgb2bgr)
            gt_img = tensor2img(visuals['gt'], rgb2bgr=rgb2bgr) if 'gt' in visuals else None
            if with_metrics:
                for name, opt_ in deepcopy(self.opt['val']['metrics']).items():
                    fn = getattr(metric_module, opt_.pop('type'))
                    a, b = (sr_img, gt_img) if use_image else (visuals['result'], visuals['gt'])
                    self.metric_results[name] += fn(a, b, **opt_)
            cnt += 1
        current_metric = 0.
        if with_metrics:
            for m in self.metric_results:
                self.metric_results[m] /= max(cnt, 1)
                current_metric = self.metric_results[m]
            logger.info('Validation ' + ', '.join(f'{m}: {v:.4f}' for m, v in self.metric_results.items()))
        return current_metric

    def get_current_visuals(self):
        out = OrderedDict(lq=self.lq.detach().cpu(), result=self.output.detach().cpu())
        if hasattr(self, 'gt'):
            out['gt'] = self.gt.detach().cpu()
        return out

    # ---- what the reference's `.state` file does not know about: the loss scale of the fp16-split backward (device-resident
    # StepGuard), the shift the range survey applied to it, and the one-way "masa" freeze (R3).  Extra key 'tdr' in the file;
    # the reference's resume_training ignores unknown keys, ours restores them.
    def extra_training_state(self):
        st = {'masa_frozen': bool(getattr(self, '_masa_frozen', False)), 'scale_shift': int(getattr(self, '_scale_shift', 0)),
              'bwd_full_range': bool(getattr(self, '_bwd_full_range', False))}
        g = getattr(self.optimizer_g, 'guard', None)
        if g is not None:
            r = g.read()
            st.update(guard_scale=float(r.scale), guard_max_scale=float(r.max_scale), skipped=int(r.skipped))
        return st

    def load_extra_training_state(self, st):
        if not st:
            return
        self._masa_frozen = bool(st.get('masa_frozen', False))
        self._scale_shift = int(st.get('scale_shift', 0))
        self._bwd_full_range = bool(st.get('bwd_full_range', False))
        if 'guard_scale' in st and isinstance(self.optimizer_g, FusedClipAdamW):
            g = self.optimizer_g.ensure_guard(self.device)
            g.write(scale=st['guard_scale'], max_scale=st.get('guard_max_scale', st['guard_scale']))

    def save(self, epoch, current_iter):
        if self.ema_decay > 0:
            self.save_network([self.net_g, self.net_g_ema], 'net_g', current_iter, param_key=['params', 'params_ema'])
        else:
            self.save_network(self.net_g, 'net_g', current_iter)
        self.save_training_state(epoch, current_iter)
