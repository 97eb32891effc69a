#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s (ndims x nparticles x L per sampling_iteration) of the
MJHMC hot path on MI355X, with the HBM roofline of the jump kernel and the NumPy CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4|c1]

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N bench.py ...
(one rank per GPU).  Particle columns are independent chains, so each rank owns its own block of
columns (global particle ids keep the RNG streams identical to an unsharded run); nothing is
exchanged inside the timed region -> weak scaling, value = all ranks' particle-steps / max time.

A "step" is one MarkovJumpHMC.sampling_iteration over all particles.
Inputs are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)

WORKLOADS = {
    # BASELINE.json configs[1]
    'c2': dict(name='C2 isotropic Gaussian ndims=512 nparticles=100000/GPU L=10 fp64', kind='iso', D=512, N=100000,
               L=10, eps=0.05, beta=0.1, dtype='float64', params=[1.0]),
    # BASELINE.json configs[3] (per-GPU share of 1e6 particles at 8 GPUs is 125000; single GPU runs all 1e6)
    'c4': dict(name='C4 Neal funnel ndims=32 nparticles=1000000/GPU L=15 fp64', kind='funnel', D=32, N=1000000,
               L=15, eps=0.05, beta=0.1, dtype='float64', params=[3.0]),
    # BASELINE.json configs[2]: ProductOfT on the matrix cores (distr_data/dump_512.pkl is absent from the
    # reference checkout -> weights from the reference's init_weights recipe)
    'c3': dict(name='C3 ProductOfT ndims=nbasis=512 nparticles=100000/GPU L=20 fp32', kind='pot', D=512, N=100000,
               L=20, eps=0.05, beta=0.1, dtype='float32', params=None),
    # BASELINE.json configs[4]: 200000 particles over 8 GPUs = 25000 per GPU; synthetic dictionary (the
    # reference's distr_data/dump_1024.pkl is not in its checkout)
    'c5': dict(name='C5 SparseImageCode n_coeffs=1024 img=256 nparticles=25000/GPU L=25 bf16 state / fp32 accumulate',
               kind='sic', D=1024, N=25000, L=25, eps=0.05, beta=0.1, dtype='bfloat16', params=None),
    # BASELINE.json configs[0] (README shape; plumbing)
    'c1': dict(name='C1 README isotropic Gaussian ndims=2 nparticles=100 L=5', kind='iso', D=2, N=100, L=5,
               eps=0.1, beta=0.1, dtype='float64', params=[1.0]),
}


def pot_model(D):
    """init_weights of mjhmc/search/MJHMC_poe_36/mjhmc_objective.py:15-23 (seed 2015) plus I (invertible)."""
    rs = np.random.RandomState(2015)
    sp_var = rs.rand(D, D)
    W = rs.randn(D, D)
    W[sp_var > 0.05] = 0
    lognu = np.log(rs.rand(D) * 2 + 2.1)
    return W + np.eye(D), lognu


def sic_model():
    """column-normalised random dictionary (256, 1024), patch y = B a0 + 0.1 noise, a0 5 % sparse"""
    rs = np.random.RandomState(0)
    B = rs.randn(256, 1024)
    B /= np.linalg.norm(B, axis=0, keepdims=True)
    a0 = rs.randn(1024) * (rs.rand(1024) < 0.05)
    y = B.dot(a0) + 0.1 * rs.randn(256)
    return B, y, a0


def initial_state(w, rank):
    rng = np.random.RandomState(1000 + rank)
    if w['kind'] == 'sic':
        return sic_model()[2][:, None] + 0.1 * rng.randn(w['D'], w['N'])
    if w['kind'] == 'pot':
        # gen_init_X of ProductOfT (distributions.py:437-445): Student-t draws mapped through inv(W)
        W, lognu = pot_model(w['D'])
        Z = np.stack([rng.standard_t(np.float32(np.exp(lognu[i])), size=w['N']) for i in range(w['D'])])
        return np.linalg.solve(W.astype(np.float32).astype(np.float64), Z)
    X0 = rng.randn(w['D'], w['N'])
    if w['kind'] == 'funnel':
        X0[0] *= w['params'][0]
        X0[1:] *= np.exp(X0[0] / 2.)
    return X0


def algorithmic_bytes_per_particle(D, esize):
    """One sampling_iteration: read X,V + write X',V' (4*D*s) + per-particle scalars:
    read EX,EV,H_flf (3s); write EX,EV,H_flf (3s) + dwell (8) + dwell-ring slot (8) + trans (1)."""
    return 4 * D * esize + 6 * esize + 17


def cpu_baseline(w, seconds_target=15.0):
    """The NumPy oracle (structurally faithful port of the reference's NumPy path) timed on this
    host: bounded column sample of the same workload."""
    from oracle import mjhmc_oracle as orc
    n = min(w['N'], 4000 if w['D'] >= 256 else 20000)
    rng = np.random.RandomState(7)
    threads = 1
    if w['kind'] in ('pot', 'sic'):
        if w['kind'] == 'pot':
            W, lognu = pot_model(w['D'])
            en = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float32)   # float32 force, float64 state: as the reference
        else:
            B, y, _ = sic_model()
            en = orc.SparseImageCode(B, y.reshape(1, -1), lmbda=0.01, cauchy=True)
        X0 = initial_state(dict(w, N=n), 7)
        try:
            from threadpoolctl import threadpool_info
            threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])   # BLAS threads used by np.dot
        except Exception:
            threads = os.cpu_count()
    else:
        X0 = rng.randn(w['D'], n)
        en = orc.IsoGaussian(w['params'][0]) if w['kind'] == 'iso' else orc.FunnelNeal(w['params'][0])
    if w['kind'] == 'funnel':
        X0[0] *= w['params'][0]
        X0[1:] *= np.exp(X0[0] / 2.)
    np.random.seed(11)
    s = orc.MarkovJumpHMC(en, X0, epsilon=w['eps'], beta=w['beta'], num_leapfrog_steps=w['L'], resample=False)
    s.sampling_iteration()                     # warm-up (all-cold first iteration)
    iters, t0 = 0, time.perf_counter()
    while True:
        s.sampling_iteration()
        iters += 1
        dt = time.perf_counter() - t0
        if dt > seconds_target or iters >= 50:
            break
    value = w['D'] * n * w['L'] * iters / dt
    return dict(value=value, unit='particle-steps/s', cores=threads, kind='port',
                sample='NumPy oracle (port of the reference path), ndims=%d, nparticles=%d of %d, L=%d, %d '
                       'sampling_iterations after 1 warm-up, %.1f s, numpy %s, os.cpu_count=%d'
                       % (w['D'], n, w['N'], w['L'], iters, dt, np.__version__, os.cpu_count()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64)     # one fused launch of the elementwise kernels
    ap.add_argument('--warmup', type=int, default=128)   # the chip needs ~15 ms of work to reach its sustained clocks
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # one rank per GPU over RCCL.  MJHMC_BENCH_BACKEND=gloo + MJHMC_BENCH_ONE_GPU=1 exist only to exercise
        # this code path with several ranks on a single-GPU box.
        backend = os.environ.get('MJHMC_BENCH_BACKEND', 'nccl')
        if os.environ.get('MJHMC_BENCH_ONE_GPU'):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from mjhmc_amd import engine, _lib
    ctx = engine.context(local_rank)
    kind = {'iso': _lib.E_ISO_GAUSS, 'funnel': _lib.E_FUNNEL_NEAL, 'pot': _lib.E_PRODUCT_OF_T,
            'sic': _lib.E_SPARSE_CODE}[w['kind']]
    params = w['params']
    if w['kind'] == 'pot':
        W, lognu = pot_model(w['D'])
        params = np.concatenate([[float(w['D'])], W.ravel(), np.exp(lognu), np.zeros(w['D'])])
    if w['kind'] == 'sic':
        B, y, _ = sic_model()
        params = np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y])
    en = engine.DeviceEnergy(ctx, kind, w['D'], params)
    X0 = initial_state(w, rank)
    smp = engine.DeviceSampler(en, X0, seed=20261002, first_particle_id=rank * w['N'], dtype=w['dtype'])
    del X0
    p_r = -np.log(1 - w['beta']) * 0.5
    smp.set_hparams(w['eps'], w['L'], p_r, 1.0)

    def barrier():
        smp.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        smp.iterate(args.warmup)
    barrier()
    t0 = time.perf_counter()
    stats, done = smp.iterate(args.steps)       # K sampling_iterations back to back, one host sync at the end
    smp.sync()
    t1 = time.perf_counter()
    barrier()
    assert done == args.steps, 'a non-finite rate interrupted the timed region'
    elapsed = t1 - t0
    tim = smp.last_timing()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same workload with one sampling iteration per launch (the HBM-bound form of the kernel), measured after the
    # timed region so that the classical roofline fraction is in the line too
    unfused_ms = None
    if w['kind'] in ('iso', 'diag') and args.steps >= 2 and not os.environ.get('MJHMC_NO_FUSE'):
        os.environ['MJHMC_NO_FUSE'] = '1'
        try:
            smp.iterate(32)
            smp.iterate(32)
            t_u = smp.last_timing()
            unfused_ms = t_u['jump_kernel_ms'] / max(t_u['n_jump_launches'], 1)
        finally:
            del os.environ['MJHMC_NO_FUSE']

    gather_info = None
    if dist is not None:
        # the one collective of the path: all-gather of sample columns at the end of sample()
        # (RCCL over xGMI).  Outside the timed region; bounded to 8192 columns per rank.
        try:
            from mjhmc_amd.parallel import Comm, ShardPlan, gather_state_columns
            comm = Comm()
            smp.ring_alloc(1)
            smp.iterate(1, ring_slot0=0)
            ncol = min(8192, w['N'])
            cols = smp.ring_gather(np.arange(ncol, dtype=np.int64))
            tg0 = time.perf_counter()
            full = gather_state_columns(comm, ShardPlan(ncol * world, world), cols)
            tg1 = time.perf_counter()
            gather_info = {'ok': bool(full.shape == (w['D'], ncol * world) and np.array_equal(full[:, rank * ncol:(rank + 1) * ncol], cols)),
                           'backend': comm.backend, 'columns_per_rank': ncol, 'ms': (tg1 - tg0) * 1e3}
        except Exception as exc:  # the bench line must survive a collective problem
            gather_info = {'ok': False, 'error': repr(exc)[:300]}

    if rank == 0:
        esize = 8 if w['dtype'] == 'float64' else 4
        units = w['D'] * w['N'] * w['L'] * args.steps * world
        kern_ms = tim['jump_kernel_ms'] / max(tim['n_jump_launches'], 1)        # per sampling iteration
        abytes = algorithmic_bytes_per_particle(w['D'], esize) * w['N']           # per sampling iteration
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        # elementwise energies: one launch covers up to 64 fused iterations (state stays on chip in between)
        fused = w['kind'] in ('iso', 'diag') and args.steps >= 2 and not os.environ.get('MJHMC_NO_FUSE')
        n_launch = -(-args.steps // 64) if fused else args.steps
        it_per_launch = args.steps / float(n_launch)
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tfile):
            # PMC-measured HBM bytes of one launch (tools/reduce_pmc.py); only quoted when it was measured on
            # launches of the same shape as the ones just timed
            rec = json.load(open(tfile)).get(args.workload)
            if isinstance(rec, dict) and abs(rec.get('iterations_per_launch', 0) - it_per_launch) < 1e-9:
                traffic = rec['bytes_per_launch']
        n_l = sum(s.l for s in stats)
        n_cold = sum(s.n_cold for s in stats)
        if w['kind'] in ('pot', 'sic'):
            # dense energy: the bound is the matrix pipe (fp32 for ProductOfT, bf16 for SparseImageCode).
            # Algorithmic flops from the exact counters: dEdX_evals * 4*D*K + E_evals * 2*D*K
            DK = float(w['D']) * (w['D'] if w['kind'] == 'pot' else 256)
            peak = 157.3 if w['kind'] == 'pot' else 2500.0
            flops = sum(s.dEdX_evals * 4 * DK + s.E_evals * 2 * DK for s in stats) / len(stats)
            achieved_tf = flops / (kern_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'achieved': achieved_tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved_tf / peak,
                    'traffic': None, 'kernel': 'pot_jump_kernel' if w['kind'] == 'pot' else 'sic_jump_kernel',
                    'avg_launch_ms': kern_ms,
                    'launches_timed': tim['n_jump_launches'], 'algorithmic_flops_per_launch': flops}
        else:
            # Vector work executed per iteration: one fused multiply-add per element for the opening half kick,
            # two per element and leapfrog step (drift, merged kick), on the forward trajectory of every particle
            # and the inverse one of the cold-cache particles, plus 4 flop per element for the two energy sums.
            vflops = (1.0 + n_cold / float(w['N'] * args.steps)) * (4.0 * w['L'] + 6.0) * w['D'] * w['N']
            valu_tf = vflops / (kern_ms * 1e-3) / 1e12
            valu_peak = 78.6 if w['dtype'] == 'float64' else 157.3
            roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                    'kernel': 'mjhmc_jump_kernel' if fused or w['N'] < 16384 or w['D'] * esize >= 2048 else
                              'mjhmc_jump_kernel + the compacted passes of one iteration (cold list, inverse-L, R list, refresh)',
                    'avg_launch_ms': kern_ms * it_per_launch,
                    'launches_timed': n_launch, 'iterations_per_launch': it_per_launch,
                    'algorithmic_bytes_per_launch': abytes * it_per_launch,
                    'note': ('fused launch: the state crosses HBM once per launch, not once per iteration, so the '
                             'algorithmic rate is not limited by HBM; the limiter is the fp64 vector pipe (see valu)')
                            if fused else 'one sampling iteration per launch',
                    'valu': {'achieved': valu_tf, 'peak': valu_peak, 'unit': 'TFLOP/s', 'frac': valu_tf / valu_peak,
                             'dtype': w['dtype'], 'what': 'trajectory + energy flops only (rates, draws, reductions '
                                                          'and bookkeeping are vector instructions too, not flops)'}}
            if unfused_ms:
                roof['one_iteration_per_launch'] = {
                    'avg_launch_ms': unfused_ms, 'achieved': abytes / (unfused_ms * 1e-3) / 1e9, 'unit': 'GB/s',
                    'frac': abytes / (unfused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'what': 'the same kernel with the state crossing HBM every iteration (MJHMC_NO_FUSE=1): HBM-bound'}
        out = {
            'metric': 'particle-steps/sec (ndims x nparticles x L)',
            'value': units / elapsed,
            'unit': 'particle-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'float64': 'f64', 'float32': 'f32', 'bfloat16': 'bf16 state / f32 accumulate'}[w['dtype']],
            'data': 'synthetic',
            'config': {'workload': w['name'], 'ndims': w['D'], 'nparticles_per_gpu': w['N'], 'L': w['L'],
                       'epsilon': w['eps'], 'beta': w['beta'], 'rng': 'philox4x32-10',
                       'particles_x_L_per_s': w['N'] * w['L'] * args.steps * world / elapsed,
                       'L_move_fraction': n_l / float(w['N'] * args.steps),
                       'cold_fraction': n_cold / float(w['N'] * args.steps)},
            'roofline': roof,
        }
        if gather_info is not None:
            out['config']['sample_gather'] = gather_info
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(w)
            out['config']['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
