import sys, time, torch
sys.path.insert(0, '/root/repo')
from oracle import nafnet_ref_oracle as O
cfg = O.default_cfg(width=32, nf=32, enc_blk_nums=[1, 1, 1, 28], ext_n_blocks=[4, 4, 4, 4], reffusion_n_blocks=[2, 2, 2, 2, 2])
P = O.synth_params(cfg, seed=0)
lq, gt, ref = O.synth_pair(1, 512, 512, seed=83)
for t in (16, 32, 64, 128):
    torch.set_num_threads(t)
    Pr = {k: v.clone().double().requires_grad_(True) for k, v in P.items()}
    t0 = time.time()
    ro = O.nafnet_ref_forward(Pr, cfg, lq.double(), ref.double())
    O.l1_loss(ro, gt.double()).backward()
    print(t, 'threads:', round(time.time() - t0, 1), 's', flush=True)
