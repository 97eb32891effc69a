#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s (ndims x nparticles x L per sampling_iteration) of the MJHMC hot path on
MI355X, each workload with the roofline that bounds its kernel and the NumPy CPU baseline timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--head c3|c2|c4|c5] [--workload c2|c3|c4|c5|c1|all]
                    [--scaling weak|strong]

The ONE JSON line rank 0 prints has the workload BASELINE.json's numeric target is stated on (C3, ProductOfT
512 x 100 000: ">= 50x the NumPy reference at 1 GPU"; in the reference's arithmetic: c3f64) as its top-level workload
(--head) and every GPU workload (C2..C5 = configs[1..4]; c5 = SparseImageCode with the reference's float32 state rows,
c5bf16 = BASELINE's wording, which runs hot under MarkovJumpHMC) as flat scalars of `config`: cK_ms, cK_frac (the flops
or bytes the chain NEEDS / time / peak -- one meaning for every workload), cK_frac_executed, cK_frac_counted, cK_bound,
cK_one_iter_ms/_frac/_call_ms (one sampling_iteration per call: launches by HIP events; the whole call by the wall clock), cK_shard{2,4,8}_{ms,frac,eff}, cK_sample<n>_incl_download,
cK_autocor_on_device; the complete records go to --detail.

A "step" is one MarkovJumpHMC.sampling_iteration over all particles; --steps K is the batch one
mjhmc_iterate call runs back to back with a single host sync.  The timed region is R such calls (R chosen so that it
lasts >= 0.5 s: a 3 ms region measures the clock ramp, not the kernel), every call bracketed by barrier + stream sync
on both sides, MAX over ranks; value = all ranks' particle-steps / the summed time, median and spread over the R
calls reported next to it.  Inputs are resident in HBM before the timed region starts.

N > 1: one rank per GPU.  Either  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...  (the ranks
find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment and pass the RCCL id through the launcher's store)
or plainly  python bench.py --gpus N ...  : with WORLD_SIZE unset the process only SPAWNS the N ranks (subprocess, before
anything touches a GPU or loads the library; never exec), gives them RANK / LOCAL_RANK / WORLD_SIZE and a rendezvous file
in a fresh directory (MJHMC_COMM_ID_FILE), relays rank 0's line and exits non-zero if any rank does.  Particle columns are
independent chains: each rank owns a block of columns (global particle ids keep the RNG streams those of an unsharded
run), nothing is exchanged inside the timed region.  --scaling weak (default): every rank runs the full single-GPU
workload (that is `value`); at N > 1 the line also carries, under "strong", C4 and C5 with the workload's particle
count as the TOTAL (BASELINE.json words them that way: "1 000 000 sharded across 8", "200 000 ... on 8xMI355X").
"""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
L2_PEAK_GBS = 34500.0      # aggregate L2 bandwidth measured in the same guide (memory hierarchy, L2)
MIN_TIMED_S = 0.5

WORKLOADS = {
    # BASELINE.json configs[1]
    'c2': dict(name='C2 isotropic Gaussian ndims=512 nparticles=100000 L=10 fp64', kind='iso', D=512, N=100000,
               L=10, eps=0.05, beta=0.1, dtype='float64', params=[1.0]),
    # BASELINE.json configs[2]: ProductOfT on the matrix cores (distr_data/dump_512.pkl is absent from the
    # reference checkout -> weights from the reference's init_weights recipe)
    'c3': dict(name='C3 ProductOfT ndims=nbasis=512 nparticles=100000 L=20 fp32 (state and force; the reference '
                    'keeps a float64 state around its float32 force)', kind='pot', D=512, N=100000,
               L=20, eps=0.05, beta=0.1, dtype='float32', params=None),
    # the same workload in the reference's own arithmetic (float64 HMCState arrays around the float32 force,
    # distributions.py:408-415 + hmc_state.py:29-38): the tile kernel with the float64 state streamed through its
    # epilogue (csrc/dense_pot64.hip) -- the line's top level since round 4
    'c3f64': dict(name='C3 ProductOfT ndims=nbasis=512 nparticles=100000 L=20, float64 state around the float32 force '
                       '(the reference\'s arithmetic)', kind='pot', D=512, N=100000,
                  L=20, eps=0.05, beta=0.1, dtype='float64', params=None),
    # BASELINE.json configs[3]
    'c4': dict(name='C4 Neal funnel ndims=32 nparticles=1000000 L=15 fp64', kind='funnel', D=32, N=1000000,
               L=15, eps=0.05, beta=0.1, dtype='float64', params=[3.0]),
    # BASELINE.json configs[4]: all 200000 particles fit one GPU; synthetic dictionary (the reference's
    # distr_data/dump_1024.pkl is not in its checkout).  The C5 OF RECORD keeps the state rows as the reference keeps them
    # (TensorFlow float32 placeholders, tf_distributions.py:89) with bf16 matrix-core operands and fp32 accumulation: under
    # MarkovJumpHMC this form keeps the target law (tests/test_gpu_stationary.py::test_sic_stationary_law, within the 1-3 %
    # the bf16 OPERANDS leave in the temperature) -- it is the class's default state_dtype
    'c5': dict(name='C5 SparseImageCode n_coeffs=1024 img=256 nparticles=200000 L=25, float32 state (the reference\'s) / bf16 '
                    'matrix-core operands / fp32 accumulate', kind='sic', D=1024, N=200000, L=25, eps=0.05, beta=0.1,
               dtype='float32', params=None),
    # BASELINE.json's wording of configs[4], "bf16 state / fp32 accumulate": parity-green against the oracle with the same
    # rounding, and statistically HOT under MarkovJumpHMC -- rounding the state at every commit makes L irreversible and the
    # jump process balances its rates on H(FLF(Lz)) = H(z) (test_sic_stationary_law asserts the drift).  Kept
    # as a labelled variant (`c5bf16_law` = "hot" in the line); same kernel, 2 % less time (HBM is at 0.28 TB/s either way)
    'c5bf16': dict(name='C5 SparseImageCode n_coeffs=1024 img=256 nparticles=200000 L=25 bf16 state / fp32 accumulate '
                        '(BASELINE.json\'s wording; the MarkovJumpHMC chain runs hot in this form)',
                   kind='sic', D=1024, N=200000, L=25, eps=0.05, beta=0.1, dtype='bfloat16', params=None),
    # BASELINE.json configs[0] (README shape; plumbing)
    'c1': dict(name='C1 README isotropic Gaussian ndims=2 nparticles=100 L=5', kind='iso', D=2, N=100, L=5,
               eps=0.1, beta=0.1, dtype='float64', params=[1.0]),
}
DTYPE_TAG = {'float64': 'f64 state (ProductOfT: around the f32 matrix-core force)', 'float32': 'f32 state and force (the reference: f64 state arrays around its f32 Theano force)',
             'bfloat16': 'bf16 state / f32 accumulate'}
DTYPE_TAG_SIC32 = 'f32 state / bf16 matrix-core operands / f32 accumulate'


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


def initial_state(w, n, stream):
    rng = np.random.RandomState(1000 + stream)
    if w['kind'] == 'sic':
        return (sic_model()[2][:, None] + 0.1 * rng.randn(w['D'], n)).astype(np.float64)
    if w['kind'] == 'pot':
        # gen_init_X of ProductOfT (distributions.py:437-445): Student-t draws mapped through inv(W)
        W, lognu = pot_model(w['D'])
        Z = np.stack([rng.standard_t(np.float32(np.exp(lognu[i])), size=n) for i in range(w['D'])])
        return np.linalg.solve(W.astype(np.float32).astype(np.float64), Z)
    X0 = rng.randn(w['D'], n)
    if w['kind'] == 'funnel':
        X0[0] *= w['params'][0]
        X0[1:] *= np.exp(X0[0] / 2.)
    return X0


def algorithmic_bytes_per_particle(D, esize):
    """One sampling_iteration of an elementwise energy: read X,V + write X',V' (4*D*s) + per-particle
    scalars: read EX,EV,H_flf (3s); write EX,EV,H_flf (3s) + dwell (8) + dwell-ring slot (8) + trans (1)."""
    return 4 * D * esize + 6 * esize + 17


CPU_BASELINE_N = 8000      # "time at N = 8 000 and state N explicitly", >= 2 timed iterations after 1 warm-up


def cpu_baseline(w, seconds_target):
    """The NumPy oracle (structurally faithful port of the reference's NumPy path) timed on this host: a bounded
    column sample of the same workload -- N = 8 000 columns (all of them for a smaller workload), one warm-up
    iteration, then at least TWO timed sampling_iterations and as many more as fit `seconds_target`."""
    from oracle import mjhmc_oracle as orc
    n = min(w['N'], CPU_BASELINE_N)
    threads = 1
    if w['kind'] in ('pot', 'sic'):
        if w['kind'] == 'pot':
            W, lognu = pot_model(w['D'])
            en = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float32)   # float32 force, float64 state: as the reference
        else:
            B, y, _ = sic_model()
            en = orc.SparseImageCode(B, y.reshape(1, -1), lmbda=0.01, cauchy=True)
        try:
            from threadpoolctl import threadpool_info
            threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])   # BLAS threads used by np.dot
        except Exception:
            threads = os.cpu_count()
    else:
        en = orc.IsoGaussian(w['params'][0]) if w['kind'] == 'iso' else orc.FunnelNeal(w['params'][0])
    X0 = initial_state(w, n, 7)
    np.random.seed(11)
    s = orc.MarkovJumpHMC(en, X0, epsilon=w['eps'], beta=w['beta'], num_leapfrog_steps=w['L'], resample=False)
    s.sampling_iteration()                     # warm-up (all-cold first iteration)
    iters, t0 = 0, time.perf_counter()
    while True:
        s.sampling_iteration()
        iters += 1
        dt = time.perf_counter() - t0
        if iters >= 2 and (dt > seconds_target or iters >= 50):
            break
    value = w['D'] * n * w['L'] * iters / dt
    return dict(value=value, unit='particle-steps/s', cores=threads, kind='port',
                sample='NumPy oracle (port of the reference path), ndims=%d, nparticles=%d of %d, L=%d, %d '
                       'sampling_iterations after 1 warm-up, %.1f s, numpy %s, os.cpu_count=%d'
                       % (w['D'], n, w['N'], w['L'], iters, dt, np.__version__, os.cpu_count()),
                nparticles=n, iterations=iters, seconds=dt)


def measured_traffic(key, it_per_launch):
    """PMC-measured HBM bytes of one launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, reduced
    by tools/reduce_pmc.py into profiles/hbm_traffic.json).  A fused launch moves the state across HBM once whatever
    its iteration count (that is what fusing means: X, V, EX, EV, H_flf stay on chip between iterations), so its
    figure is quoted for any number of fused iterations; everything else must match the launch shape."""
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if not os.path.exists(path):
        return None
    rec = json.load(open(path)).get(key)
    if not isinstance(rec, dict):
        return None
    if rec.get('fused') or abs(rec.get('iterations_per_launch', 0) - it_per_launch) < 1e-9:
        return rec['bytes_per_launch']
    return None


class StoreRendezvous(object):
    """put / get / all_agree (the interface of mjhmc_amd.parallel.FileRendezvous) over the launcher's key/value store:
    under torch.distributed.run every rank finds MASTER_ADDR / MASTER_PORT in its environment and the elastic agent
    already serves a TCPStore there (TORCHELASTIC_USE_AGENT_STORE); without an agent rank 0 serves it."""

    def __init__(self, rank, world):
        from datetime import timedelta
        from torch.distributed import TCPStore
        agent = os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
        self.rank, self.world = rank, world
        self.store = TCPStore(host_name=os.environ['MASTER_ADDR'], port=int(os.environ['MASTER_PORT']), world_size=None,
                              is_master=(rank == 0) and not agent, timeout=timedelta(seconds=300), wait_for_workers=False)
        self.prefix = 'mjhmc_bench/%s/' % os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')

    def put(self, key, data):
        self.store.set(self.prefix + key, bytes(data))

    def get(self, key):
        return bytes(self.store.get(self.prefix + key))          # blocks until the key exists

    def all_agree(self, name, ok):
        self.put('%s.%d' % (name, self.rank), b'1' if ok else b'0')
        return all(self.get('%s.%d' % (name, r)) == b'1' for r in range(self.world))


class DeviceMonitor(object):
    """Clock, socket power and busy % of THIS rank's GPU, sampled from sysfs (amdgpu hwmon: freq1_input = sclk in Hz,
    power1_input / power1_average in microwatts; gpu_busy_percent) by a thread while a timed region runs -- the line
    then carries its own device-side evidence, and a fraction of peak can be read against the clock actually held.
    The device is found by the PCI address the library reports (mjhmc_ctx_info)."""

    def __init__(self, ctx):
        self.dir, self.hw = None, None
        try:
            m = re.search(r'\[([0-9a-fA-F:.]+)\]', ctx.info()['name'])
            cand = []
            if m:
                cand.append('/sys/bus/pci/devices/%s' % m.group(1).lower())
            for c in cand:
                if os.path.exists(os.path.join(c, 'gpu_busy_percent')):
                    self.dir = c
                    break
            if self.dir:
                hw = os.path.join(self.dir, 'hwmon')
                for h in sorted(os.listdir(hw)):
                    if os.path.exists(os.path.join(hw, h, 'freq1_input')):
                        self.hw = os.path.join(hw, h)
                        break
        except Exception:
            self.dir = None
        self.rows, self._stop, self._th = [], None, None

    @staticmethod
    def _num(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _sample(self):
        sclk = self._num(os.path.join(self.hw, 'freq1_input')) if self.hw else None
        pw = None
        if self.hw:
            pw = self._num(os.path.join(self.hw, 'power1_input'))
            if pw is None:
                pw = self._num(os.path.join(self.hw, 'power1_average'))
        busy = self._num(os.path.join(self.dir, 'gpu_busy_percent'))
        return (sclk / 1e6 if sclk else None, pw / 1e6 if pw else None, busy)

    def start(self):
        self.rows = []
        if not self.dir:
            return
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                self.rows.append(self._sample())
                self._stop.wait(0.02)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        if self._th is None:
            return {'available': False, 'why': 'no sysfs entry for this device'}
        self._stop.set()
        self._th.join()
        self._th = None
        out = {'available': True, 'samples': len(self.rows), 'source': 'sysfs hwmon (freq1_input, power1_input) + gpu_busy_percent, every 20 ms of the timed region'}
        for k, name in enumerate(('sclk_mhz', 'power_w', 'busy_pct')):
            v = [r[k] for r in self.rows if r[k] is not None]
            out[name] = [float(np.min(v)), float(np.median(v)), float(np.max(v))] if v else None
        return out


class Rig(object):
    """process-wide plumbing of one bench run: one rank per GPU; the ranks meet through the library's own RCCL
    communicator (mjhmc_comm_*, mjhmc_amd/parallel.py: RcclComm) -- barrier, MAX of the elapsed times, and the sample
    all-gather checked after the timed regions.  The 128-byte RCCL id travels through an explicit channel: the file the
    spawning parent names in MJHMC_COMM_ID_FILE (a fresh directory), or the launcher's store under
    torch.distributed.run.  Whether RCCL is used at all is decided by ALL ranks before any of them enters a collective
    (a flag per rank through the same channel): if any rank cannot load librccl, or the communicator does not come up
    everywhere, every rank takes its barrier / MAX through torch.distributed gloo instead and the line says so
    (config.comm_note).  MJHMC_BENCH_BACKEND=gloo + MJHMC_BENCH_ONE_GPU=1 exist only to exercise the multi-rank code
    path with several ranks on a single-GPU box (RCCL refuses two ranks on one device)."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        if self.world != args.gpus:
            raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d in the environment (launch either as `python bench.py '
                             '--gpus N` with WORLD_SIZE unset, or under torch.distributed.run --nproc-per-node N)'
                             % (args.gpus, self.world))
        if os.environ.get('MJHMC_BENCH_ONE_GPU'):
            self.local_rank = 0
        from mjhmc_amd import engine
        self.ctx = engine.context(self.local_rank)
        self.monitor = DeviceMonitor(self.ctx)
        progress('ctx_ready')                      # library loaded, device context up: from here on a stall is a hang
        self.comm = None
        self.comm_note = None
        if self.world > 1:
            # RCCL (and gloo) print a banner on stdout when a communicator comes up; stdout carries the ONE JSON line
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            try:
                self._connect()
            finally:
                sys.stdout.flush()
                os.dup2(keep, 1)
                os.close(keep)

    def _rendezvous(self):
        from mjhmc_amd.parallel import FileRendezvous
        path = os.environ.get('MJHMC_COMM_ID_FILE')
        if path:
            return FileRendezvous(os.path.dirname(path) or '.', self.rank, self.world)
        return StoreRendezvous(self.rank, self.world)

    def _gloo(self, note):
        if os.environ.get('MJHMC_BENCH_NOTE'):
            note = os.environ['MJHMC_BENCH_NOTE']
        self.comm_note = note
        sys.stderr.write('rank %d: %s\n' % (self.rank, note))
        import torch.distributed as dist
        from mjhmc_amd.parallel import Comm
        dist.init_process_group('gloo')
        self.comm = Comm()

    def _connect(self):
        from mjhmc_amd import _lib
        from mjhmc_amd.parallel import RcclComm
        rdv = self._rendezvous()
        want_rccl = os.environ.get('MJHMC_BENCH_BACKEND', 'rccl') != 'gloo'
        loadable = want_rccl and _lib.load().mjhmc_comm_available() == 0
        if not rdv.all_agree('rccl_loadable', loadable):         # decided by everybody BEFORE anybody enters a collective
            return self._gloo('gloo requested (MJHMC_BENCH_BACKEND)' if not want_rccl else
                              'librccl not loadable on every rank; barrier / MAX through torch.distributed gloo')
        comm, err = None, None
        try:
            comm = RcclComm(self.rank, self.world, device=self.local_rank, rendezvous=rdv)
        except Exception as exc:
            err = repr(exc)[:200]
        if rdv.all_agree('rccl_up', comm is not None):
            self.comm = comm
            return
        if comm is not None:
            comm.close()
        self._gloo('RCCL communicator did not come up on every rank (%s); barrier / MAX through torch.distributed gloo' % err)

    def barrier(self, smp):
        smp.sync()
        if self.comm is not None:
            self.comm.barrier()

    def max_over_ranks(self, values):
        if self.comm is None:
            return list(values)
        return [float(v) for v in self.comm.allreduce_f64(np.asarray(list(values), dtype=np.float64), 'max')]


def run_workload(rig, key, steps, warmup, cpu_seconds, scaling='weak', shard_of=1):
    """shard_of = G > 1: the workload at N/G particles on this one GPU -- what each rank of a G-GPU strong-scaled run
    executes (no timed region holds a collective, so t(G GPUs, N) = t(1 GPU, N/G)); same kernels, same flags."""
    from mjhmc_amd import engine, _lib
    rank, world = rig.rank, rig.world
    w = dict(WORKLOADS[key])
    n_total = w['N'] if scaling == 'strong' else w['N'] * world
    n_rank = n_total // world // shard_of
    first = rank * n_rank
    kind = {'iso': _lib.E_ISO_GAUSS, 'funnel': _lib.E_FUNNEL_NEAL, 'pot': _lib.E_PRODUCT_OF_T,
            'sic': _lib.E_SPARSE_CODE}[w['kind']]
    params = w['params']
    if w['kind'] == 'pot':
        W, lognu = pot_model(w['D'])
        params = np.concatenate([[float(w['D'])], W.ravel(), np.exp(lognu), np.zeros(w['D'])])
    if w['kind'] == 'sic':
        B, y, _ = sic_model()
        params = np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y])
    en = engine.DeviceEnergy(rig.ctx, kind, w['D'], params)
    X0 = initial_state(w, n_rank, rank)
    smp = engine.DeviceSampler(en, X0, seed=20261002, first_particle_id=first, dtype=w['dtype'])
    del X0
    p_r = -np.log(1 - w['beta']) * 0.5
    smp.set_hparams(w['eps'], w['L'], p_r, 1.0)

    # warm-up: the requested W iterations, then calls of K until the chip has been busy for ~0.2 s (sustained clocks)
    if warmup > 0:
        smp.iterate(warmup)
    smp.sync()
    t0 = time.perf_counter()
    calib = []
    while time.perf_counter() - t0 < 0.2 or len(calib) < 2:
        tc = time.perf_counter()
        smp.iterate(steps)
        smp.sync()
        calib.append(time.perf_counter() - tc)
    per_call = rig.max_over_ranks([min(calib)])[0]
    reps = int(min(max(1, np.ceil(MIN_TIMED_S / per_call)), 4096))

    call_s, kern_ms, launches = [], 0.0, 0
    agg = np.zeros(7)                               # l moves, cold caches, E evaluations, dEdX evaluations, f moves, r moves, inverse-L trajectories run
    rig.monitor.start()
    for _ in range(reps):
        rig.barrier(smp)
        tb = time.perf_counter()
        stats, done = smp.iterate(steps)            # K sampling_iterations back to back, one host sync at the end
        smp.sync()
        te = time.perf_counter()
        rig.barrier(smp)
        assert done == steps, 'a non-finite rate interrupted the timed region'
        call_s.append(te - tb)
        tim = smp.last_timing()                     # HIP events on the sampler's stream around the call's launches
        kern_ms += tim['jump_kernel_ms']
        launches += tim['n_jump_launches']
        agg += [sum(s.l for s in stats), sum(s.n_cold for s in stats), sum(s.E_evals for s in stats),
                sum(s.dEdX_evals for s in stats), sum(s.f for s in stats), sum(s.r for s in stats),
                sum(s.n_flf_run for s in stats)]
    device = rig.monitor.stop()
    call_s = rig.max_over_ranks(call_s)
    elapsed = float(np.sum(call_s))
    iters = steps * reps

    # mjhmc_iterate fuses the Gaussian forces whenever a call has >= 2 iterations -- and the funnels, whose float64 rows of up
    # to 32 dims run a lane per particle (mjhmc_fused_rows_kernel: bound by the vector pipe at any batch size)
    fused = (w['kind'] == 'iso' or (w['kind'] == 'funnel' and w['dtype'] == 'float64' and 9 <= w['D'] <= 32)) and steps >= 2
    # the same workload with one sampling iteration per launch (the HBM-bound form of the kernel: what every
    # sampling_iteration() caller gets -- a call of ONE iteration is never fused), after the timed region
    unfused_ms = unfused_call_ms = None
    if fused and shard_of == 1:
        for _ in range(8):
            smp.iterate(1)
        t_sum = 0.0
        smp.sync()
        smp.set_timing(False)                           # as the drop-in classes run it: no event pair around the launches
        t_b = time.perf_counter()
        for _ in range(32):
            smp.iterate(1)                              # (returns after its read-back: a call is synchronous)
        unfused_call_ms = (time.perf_counter() - t_b) / 32 * 1e3   # what a sampling_iteration() caller waits, host side included
        smp.set_timing(True)
        for _ in range(32):
            smp.iterate(1)
            t_sum += smp.last_timing()['jump_kernel_ms']
        unfused_ms = t_sum / 32
    # The C-ABI boundary hands over HOST buffers in the reference's (ndims, nparticles) layout; the device keeps rows
    # per particle.  What that costs (re-tile kernel + PCIe, pageable host memory), outside `value`: a state read,
    # and a batch of stacked samples = n iterations into the device ring + the download of the ring (sample()'s own form:
    # every slot crosses PCIe while the following iterations run, mjhmc_iterate_download).  n = 10 where ten float64 host
    # copies of the state stay under ~4.4 GB, fewer for the bigger states (C5: 1.6 GB per sample).
    boundary = None
    want_boundary = world == 1 and scaling == 'weak' and shard_of == 1 and key in (rig.head, 'c2', 'c4', 'c5')
    if want_boundary:
        esz = {'float64': 8, 'float32': 4, 'bfloat16': 2}[w['dtype']]
        host_bytes = 8 * int(w['D']) * n_rank
        n_s = int(max(2, min(10, 4.4e9 // host_bytes)))
        smp.ring_alloc(n_s)
        ring = np.empty((w['D'], n_s * n_rank))
        ring[:] = 0.0                                              # (pages faulted in before anything is timed)
        t_b = time.perf_counter()
        smp.iterate(n_s, ring_slot0=0)
        smp.sync()
        t_it = time.perf_counter() - t_b
        smp.iterate_download(n_s, 0, ring)                         # (first call: staging, pinned buffers, stream, thread)
        t_b = time.perf_counter()
        smp.iterate_download(n_s, 0, ring)
        t_str = time.perf_counter() - t_b
        psteps = float(w['D']) * n_rank * w['L'] * n_s
        boundary = {'samples': n_s, 'host_bytes': int(ring.nbytes), 'iterate_ms': t_it * 1e3, 'streamed_ms': t_str * 1e3,
                    'particle_steps_per_s_incl_download': psteps / t_str, 'incl_download_over_resident': t_it / t_str,
                    'state_bytes_device': int(w['D']) * n_rank * esz}
        if key == rig.head:
            t_b = time.perf_counter()
            Xh = smp.read(_lib.F_X)
            t_read = time.perf_counter() - t_b
            t_b = time.perf_counter()
            smp.read(_lib.F_X, out=Xh)
            t_read2 = time.perf_counter() - t_b
            t_b = time.perf_counter()
            smp.ring_read(0, n_s, stacked=False, out=ring)         # sample(..., out=preallocated) of a recorded ring
            t_dl2 = time.perf_counter() - t_b
            boundary.update({'state_read_GBps': Xh.nbytes / t_read / 1e9, 'state_read_warm_GBps': Xh.nbytes / t_read2 / 1e9,
                             'download_into_preallocated_GBps': ring.nbytes / t_dl2 / 1e9,
                             'iterate_then_download_particle_steps_per_s': psteps / (t_it + t_dl2)})
            del Xh
        del ring
    # ... and the consumer that needs no download: the reference's main caller records samples to autocorrelate them
    # (misc/autocor.py:213-261 generate_samples, :37-49 fft_autocor).  T iterations recorded into the device ring
    # (sample(n, preserve_order=True)'s form) + the lag sums over the ring on the device (mjhmc_ring_autocor): the samples
    # never leave HBM; particle-steps/s over BOTH.
    acor = None
    if world == 1 and scaling == 'weak' and shard_of == 1 and key in ('c2', 'c4'):
        T_ac = 16
        smp.ring_alloc(T_ac)
        smp.iterate(T_ac, ring_slot0=0)
        smp.ring_autocor(0, T_ac)                                  # (first call: plans, work buffers)
        smp.sync()
        t_b = time.perf_counter()
        smp.iterate(T_ac, ring_slot0=0)
        smp.sync()
        t_rec = time.perf_counter() - t_b
        t_b = time.perf_counter()
        lag = smp.ring_autocor(0, T_ac)
        t_ac = time.perf_counter() - t_b
        acor = {'samples': T_ac, 'record_ms': t_rec * 1e3, 'autocor_ms': t_ac * 1e3, 'ring_bytes': int(w['D']) * n_rank * T_ac * 8,
                'particle_steps_per_s_incl_autocor': float(w['D']) * n_rank * w['L'] * T_ac / (t_rec + t_ac),
                'finite': bool(np.isfinite(lag).all())}
    smp.close()
    if rank != 0:
        return None

    esize = {'float64': 8, 'float32': 4, 'bfloat16': 2}[w['dtype']]
    units = float(w['D']) * n_rank * w['L'] * iters * world
    kern_it_ms = kern_ms / max(launches, 1)                       # device time per sampling iteration (HIP events)
    n_launch_call = -(-steps // 64) if fused else steps
    it_per_launch = steps / float(n_launch_call)
    cold_frac = agg[1] / float(n_rank * iters)
    full_shape = shard_of == 1 and not (scaling == 'strong' and world > 1)   # the PMC passes were taken on the single-GPU shapes
    # What every field below means (kernels, byte and flop models, why each bound) is DESIGN.md section 7 ("the line's
    # fields"); the line itself carries numbers only, so that it stays under the 8 kB the driver keeps.
    # ONE definition of `frac` for every workload: the flops (or bytes) the CHAIN NEEDS / time / peak.  Beside it
    # `executed` (what the device ran: more than needed where lanes integrate trajectories nobody reads, less where a
    # term is not evaluated separately) and, for the dense energies, `counted` (the reference's evaluation count).
    if w['kind'] in ('pot', 'sic'):
        # dense energy: the bound is the matrix pipe (fp32 for ProductOfT, bf16 for SparseImageCode).
        DK = float(w['D']) * (w['D'] if w['kind'] == 'pot' else 256)
        # ProductOfT keeps dE/dX as part of the state (read + written with X, V); SparseImageCode recomputes it
        dense_bytes = (6.0 if w['kind'] == 'pot' else 4.0) * w['D'] * esize + 6 * (8 if w['dtype'] == 'float64' else 4) + 17
        peak = 157.3 if w['kind'] == 'pot' else 2500.0
        # needed: the trajectories whose end point the chain reads -- every particle's L proposal and the inverse-L proposal
        # of every cold cache EXCEPT the F-movers' (it is the L proposal of the iteration before, bit for bit:
        # csrc/dense_pot.hip) -- at L gradients (4 D K flop) + one energy (2 D K) each.
        # executed: the same trajectories without the 2 D K energy term -- the kernels take the energy from the last
        # gradient's intermediate (dense_pot_tile.hpp), no separate product runs.
        # counted: the reference's own counters (it integrates the F-movers' proposals too): dEdX_evals 4DK + E_evals 2DK.
        run = float(n_rank * iters + agg[6])                                  # trajectories integrated (forward + inverse-L)
        flops = run * (w['L'] * 4 * DK + 2 * DK) / iters
        flops_exec = run * (w['L'] * 4 * DK) / iters
        flops_counted = (agg[3] * 4 * DK + agg[2] * 2 * DK) / iters
        sec = kern_it_ms * 1e-3
        tf = flops / sec / 1e12
        roof = {'bound': 'mfma', 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak,
                'executed': {'achieved': flops_exec / sec / 1e12, 'frac': flops_exec / sec / 1e12 / peak, 'flops_per_launch': flops_exec},
                'counted': {'achieved': flops_counted / sec / 1e12, 'frac': flops_counted / sec / 1e12 / peak,
                            'flops_per_launch': flops_counted},
                'traffic': measured_traffic(key, 1) if full_shape else None,
                'kernel': ('pot64_jump_kernel+pot64_decide_kernel' if (w['kind'] == 'pot' and w['dtype'] == 'float64') else
                           'pot_jump_kernel+pot_fix_kernel' if w['kind'] == 'pot' else 'sic_jump_kernel+sic_fix_kernel'),
                'avg_launch_ms': kern_it_ms, 'launches_timed': launches, 'algorithmic_flops_per_launch': flops,
                'hbm': {'algorithmic_bytes_per_launch': dense_bytes * n_rank,
                        'achieved': dense_bytes * n_rank / sec / 1e9, 'unit': 'GB/s'}}
        if w['kind'] == 'pot' and w['dtype'] == 'float64':
            # the float64-state kernel streams the position through a working copy once per leapfrog step:
            # 8 B read + 8 B written per element and gradient evaluation -- the kernel's own traffic, NOT part of the
            # algorithmic bytes of SURVEY 8d (PMC traffic / algorithmic = 8.6: that copy)
            roof['hbm']['working_copy_bytes_per_launch'] = 16.0 * w['D'] * run * w['L'] / iters
        if w['kind'] == 'sic':
            # one pass over the 512 KB dictionary per leapfrog step of a 32-particle tile, plus two per trajectory (the
            # residual at its head, the closing half kick): L + 2 passes per L gradient evaluations, out of L2 (the
            # dictionary cannot stay in a CU).
            tile_grads = run * w['L'] / 32.0 / iters
            l2 = tile_grads * (w['L'] + 2.0) / w['L'] * 512 * 1024 / sec / 1e9
            roof['l2'] = {'achieved': l2, 'peak': L2_PEAK_GBS, 'unit': 'GB/s', 'frac': l2 / L2_PEAK_GBS}
            # The bound the 32-column tile sets: every leapfrog step streams the whole dictionary out of L2 for
            # 4*D*K*32 flop = 64 flop per dictionary byte; ceiling = min(MFMA peak, L2 peak x 64 flop/B)
            fl_per_b = 4.0 * DK * 32 / (512 * 1024)
            ceil_tf = min(peak, L2_PEAK_GBS * fl_per_b / 1e3)
            roof['l2_stream'] = {'bound': 'l2-stream', 'flop_per_l2_byte': fl_per_b, 'achieved': tf, 'peak': ceil_tf,
                                 'unit': 'TFLOP/s', 'frac': tf / ceil_tf}
            # LDS: BOTH operands of every MFMA come out of LDS, 2 KiB per v_mfma_f32_32x32x16_bf16 -- 2 MiB of LDS reads per
            # tile and leapfrog step (1024 MFMAs) -- plus the dictionary landing (512 KiB per step) and the published
            # fragments (80 KiB), against 256 B/clk/CU at 2.4 GHz
            lds_bytes = tile_grads * (2.0 + 0.578) * 1024 * 1024
            lds_peak = 256.0 * 256 * 2.4                                  # B/clk/CU x CUs x GHz = GB/s
            roof['lds'] = {'achieved': lds_bytes / sec / 1e9, 'peak': lds_peak, 'unit': 'GB/s',
                           'frac': lds_bytes / sec / 1e9 / lds_peak}
    else:
        abytes = algorithmic_bytes_per_particle(w['D'], esize) * n_rank          # per sampling iteration
        hbm = {'achieved': abytes / (kern_it_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
               'algorithmic_bytes_per_launch': abytes * it_per_launch}
        hbm['frac_of_algorithmic'] = hbm['achieved'] / HBM_PEAK_GBS
        # Vector work the chain needs per iteration: one fused multiply-add per element for the opening half kick, two per
        # element and leapfrog step (drift, merged kick), on the forward trajectory of every particle and the
        # inverse one of the cold-cache particles, plus 4 flop per element for the two energy sums (rates, draws,
        # reductions and bookkeeping are vector instructions too, not flops).
        # (the funnel's force needs sum_k x_k^2 at every step: one more multiply-add per element and step; its exp(-x0),
        # once per particle and step, is not counted)
        fl_step = 6.0 if w['kind'] == 'funnel' else 4.0
        per_traj = (fl_step * w['L'] + 6.0) * w['D']
        vflops = (1.0 + cold_frac) * per_traj * n_rank
        valu_tf = vflops / (kern_it_ms * 1e-3) / 1e12
        valu_peak = 78.6 if w['dtype'] == 'float64' else 157.3
        valu = {'achieved': valu_tf, 'peak': valu_peak, 'unit': 'TFLOP/s', 'frac': valu_tf / valu_peak}
        traffic = measured_traffic(key, it_per_launch) if full_shape else None
        if fused:
            # the state crosses HBM once per LAUNCH, not once per iteration: HBM does not bound the launch (the
            # algorithmic byte rate of SURVEY 8d exceeds the HBM peak); the fp64 vector pipe does
            roof = dict(valu, bound='fp64_valu', traffic=traffic,
                        kernel='mjhmc_fused_rows_relay_kernel' if w['kind'] == 'funnel' else 'mjhmc_jump_kernel<FUSED>',
                        avg_launch_ms=kern_it_ms * it_per_launch, launches_timed=launches / it_per_launch,
                        iterations_per_launch=it_per_launch, algorithmic_flops_per_launch=vflops * it_per_launch)
            roof['hbm_algorithmic'] = hbm
            ex_flops = vflops       # a wavefront per particle (C2): the inverse-L trajectory runs for the cold particles only
            if w['kind'] == 'funnel':
                # the relay kernel (elementwise.hpp): a workgroup of 256 particles integrates its own 256 L proposals and, on
                # one wave's worth of lanes (two per pooled particle), the inverse-L proposals of its cold caches -- one
                # quarter of a wave-trajectory per wave and iteration whenever the pool is not empty: 1.125 executed per
                # 1 + cold_fraction needed (the one-wave kernel of round 5: 1 + 0.99)
                pool_busy = 1.0 - (1.0 - cold_frac) ** 256
                ex_flops = (1.0 + 0.125 * pool_busy) * per_traj * n_rank
                roof['pool_busy_fraction'] = pool_busy
            roof['executed'] = {'achieved': ex_flops / (kern_it_ms * 1e-3) / 1e12, 'unit': 'TFLOP/s',
                                'frac': ex_flops / (kern_it_ms * 1e-3) / 1e12 / valu_peak,
                                'flops_per_launch': ex_flops * it_per_launch, 'over_needed': ex_flops / vflops}
            if unfused_ms:
                roof['one_iteration_per_launch'] = {
                    'bound': 'hbm', 'avg_launch_ms': unfused_ms, 'achieved': abytes / (unfused_ms * 1e-3) / 1e9,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': abytes / (unfused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'traffic': measured_traffic(key + '_one_iteration_per_launch', 1),
                    'call_ms': unfused_call_ms}   # wall clock of a whole call (launches + read-back + host)
        else:
            roof = {'bound': 'hbm', 'achieved': hbm['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': hbm['achieved'] / HBM_PEAK_GBS, 'traffic': traffic,
                    'kernel': 'mjhmc_jump_kernel' if n_rank < 16384 or w['D'] * esize >= 2048 else
                              'mjhmc_jump_kernel+compacted passes',
                    'avg_launch_ms': kern_it_ms, 'launches_timed': launches, 'iterations_per_launch': 1.0,
                    'algorithmic_bytes_per_launch': abytes, 'valu': valu}
    alg_bytes = (roof.get('hbm', {}).get('algorithmic_bytes_per_launch') or roof.get('algorithmic_bytes_per_launch')
                 or roof.get('hbm_algorithmic', {}).get('algorithmic_bytes_per_launch'))
    if fused:
        alg_bytes = abytes                     # a fused launch has to move the state across HBM once, whatever it fuses
    roof['traffic_over_algorithmic'] = (roof['traffic'] / alg_bytes) if roof.get('traffic') and alg_bytes else None
    per_step = np.array(call_s) * 1e3 / steps
    moves = float(n_rank * iters)
    out = {
        'value': units / elapsed, 'unit': 'particle-steps/s', 'ms_per_step': elapsed * 1e3 / iters,
        'ms_per_step_median': float(np.median(per_step)), 'ms_per_step_min': float(per_step.min()),
        'ms_per_step_max': float(per_step.max()), 'repeats': reps, 'timed_s': elapsed,
        'dtype': DTYPE_TAG_SIC32 if (w['kind'] == 'sic' and w['dtype'] == 'float32') else DTYPE_TAG[w['dtype']],
        'config': {'workload': w['name'], 'ndims': w['D'], 'nparticles_per_gpu': n_rank, 'nparticles_total': n_rank * world,
                   'L': w['L'], 'epsilon': w['eps'], 'beta': w['beta'], 'rng': 'philox4x32-10',
                   'particles_x_L_per_s': n_rank * world * w['L'] * iters / elapsed,
                   'L_move_fraction': agg[0] / moves, 'F_move_fraction': agg[4] / moves, 'R_move_fraction': agg[5] / moves,
                   'cold_fraction': cold_frac, 'inverse_L_run_fraction': agg[6] / moves},
        'roofline': roof,
        'device': device,
    }
    if shard_of > 1:
        out['config']['shard_of'] = shard_of
    if device.get('sclk_mhz') and roof['bound'] == 'mfma':
        # the fraction again, against the peak at the clock the chip actually held (the guide's matrix / LDS peaks assume
        # 2.4 GHz; an HBM or vector-pipe figure does not scale that way and gets none)
        roof['frac_at_held_clock'] = roof['frac'] * 2400.0 / max(device['sclk_mhz'][1], 1.0)
        if 'lds' in roof:
            roof['lds']['frac_at_held_clock'] = roof['lds']['frac'] * 2400.0 / max(device['sclk_mhz'][1], 1.0)
    if boundary is not None:
        out['boundary'] = boundary
    if acor is not None:
        out['autocor_on_device'] = acor
    if world == 1 and cpu_seconds > 0:
        out['cpu_baseline'] = cpu_baseline(w, cpu_seconds)
        out['config']['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
    if w['kind'] == 'pot':
        out['config']['arithmetic'] = 'f32 state + f32 MFMA force' if w['dtype'] == 'float32' else 'f64 state around the f32 MFMA force (the reference\'s arithmetic)'
    return out


def rnd(x, sig=5):
    """numbers of the printed line: 5 significant digits (the detail file keeps everything)"""
    if isinstance(x, bool) or x is None or isinstance(x, (str, int)):
        return x
    if isinstance(x, float):
        return float('%.*g' % (sig, x)) if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: rnd(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [rnd(v, sig) for v in x]
    return rnd(float(x), sig)


def compact(rec):
    """One workload's record as the printed line carries it beside the flat keys of `config` (numbers only; the full record
    goes to --detail)."""
    r = rec['roofline']
    c = rec['config']
    out = {'value': rec['value'], 'steps': rec.get('steps'), 'n': c['nparticles_per_gpu'], 'achieved': r['achieved'], 'peak': r['peak'],
           'traffic': r.get('traffic'), 'cold': c['cold_fraction'], 'inv_run': c['inverse_L_run_fraction']}
    if 'frac_at_held_clock' in r:
        out['frac_at_held_clock'] = r['frac_at_held_clock']
    d = rec.get('device') or {}
    if d.get('sclk_mhz'):
        out['sclk_w'] = [d['sclk_mhz'][1], d['power_w'][1] if d.get('power_w') else None]
    if 'cpu_baseline' in rec:
        b = rec['cpu_baseline']
        out['cpu'] = [b['value'], b['cores'], b['nparticles'], b['iterations']]
    return out


def progress(marker):
    """One line per milestone of this rank into $MJHMC_BENCH_PROGRESS.<rank> (set by the spawning parent, which watches the
    files: a rank that stops making progress -- a hung collective cannot be cancelled from inside -- is found within a
    bounded time instead of at the end of the driver's budget)."""
    path = os.environ.get('MJHMC_BENCH_PROGRESS')
    if not path:
        return
    try:
        with open('%s.%s' % (path, os.environ.get('RANK', '0')), 'a') as f:
            f.write('%.3f %s\n' % (time.time(), marker))
    except OSError:
        pass


def fake_rank(args):
    """MJHMC_BENCH_FAKE=1 (tests/test_bench_spawn.py): a rank that touches no GPU and loads nothing -- it only walks
    through the milestones, and under MJHMC_BENCH_FAKE_HANG=<rank> that rank hangs where a broken collective would,
    unless the attempt runs on the gloo safety net.  Exercises the spawning parent's watchdog on a CPU-only host."""
    rank = int(os.environ.get('RANK', '0'))
    progress('ctx_ready')
    if os.environ.get('MJHMC_BENCH_BACKEND', 'rccl') != 'gloo' and os.environ.get('MJHMC_BENCH_FAKE_HANG') == str(rank):   # (the first attempt only)
        time.sleep(3600)
    progress('comm_up')
    progress('gather_done')
    gloo = os.environ.get('MJHMC_BENCH_BACKEND', 'rccl') == 'gloo'
    time.sleep(0.2 if gloo else float(os.environ.get('MJHMC_BENCH_FAKE_WORK_S', '0.2')))
    progress('workload_done fake')
    if rank == 0:
        print(json.dumps({'fake': True, 'n_gpus': args.gpus, 'comm_note': os.environ.get('MJHMC_BENCH_NOTE')}))
    return 0


def sample_gather_check(rig):
    """The one collective of the path: the all-gather of sample columns at the end of sample() (RCCL over xGMI, device
    ring to device ring), outside every timed region, on a small sampler with UNEVEN column shards.  Every rank also
    runs the unsharded sampler on its own GPU (the counter RNG is keyed by global particle id, so the shards ARE the
    unsharded run's columns) and compares the three gathered forms with it bit for bit: time-major blocks
    (np.concatenate(axis=1)), stacked (np.stack(axis=-1)) and the dwell-time-resampled columns."""
    try:
        from mjhmc_amd import engine, _lib
        from mjhmc_amd.parallel import ShardPlan, assemble_resample, assemble_stacked
        comm = rig.comm
        D, n = 32, 4
        total = 8192 * rig.world + max(1, rig.world // 2)     # not divisible: the first shards hold one column more
        plan = ShardPlan(total, rig.world)
        lo, hi = plan.span(rig.rank)
        X0 = np.random.RandomState(77).randn(D, total)
        en = engine.DeviceEnergy(rig.ctx, _lib.E_ISO_GAUSS, D, [1.0])

        def run(cols, first):
            smp = engine.DeviceSampler(en, X0[:, cols], seed=1, first_particle_id=first)
            smp.set_hparams(0.1, 5, 0.05, 1.0)
            smp.ring_alloc(n)
            smp.iterate(n, ring_slot0=0)
            return smp

        smp = run(slice(lo, hi), lo)
        whole = run(slice(0, total), 0)
        want_cat, want_cube = whole.ring_read(0, n, stacked=False), whole.ring_read(0, n, stacked=True)
        u = np.random.RandomState(78).rand(n * total)
        comm.barrier()
        tg0 = time.perf_counter()
        if comm.on_device:
            got_cat = comm.allgather_ring(smp, 0, n, False, plan.counts)
        else:
            got_cat = assemble_stacked(comm, plan, smp.ring_read(0, n, stacked=False), n, False)
        tg1 = time.perf_counter()
        if comm.on_device:
            got_cube = comm.allgather_ring(smp, 0, n, True, plan.counts)
        else:
            got_cube = assemble_stacked(comm, plan, smp.ring_read(0, n, stacked=True), n, True)
        got_pick, idx = assemble_resample(comm, plan, n, smp.ring_read_dwell(0, n), smp.ring_gather, uniforms=u,
                                          dev=smp if comm.on_device else None)
        want_pick = whole.ring_gather(idx)
        checks = {'time_major': bool(np.array_equal(got_cat, want_cat)), 'stacked': bool(np.array_equal(got_cube, want_cube)),
                  'resampled_columns': bool(np.array_equal(got_pick, want_pick))}
        ok_all = int(comm.allreduce_ints([1 if all(checks.values()) else 0], 'min')[0])
        info = {'ok': bool(ok_all), 'checks_rank0': checks, 'backend': comm.backend,
                'ranks': comm.count() if hasattr(comm, 'count') else comm.world,
                'columns_per_rank': [int(c) for c in plan.counts], 'slots': n, 'ms_time_major': (tg1 - tg0) * 1e3,
                'bytes_per_rank': int(D * plan.counts[0] * n * 8), 'device_to_device': bool(comm.on_device),
                'compared_with': 'the unsharded sampler run on every rank\'s own GPU, bit for bit'}
        smp.close()
        whole.close()
        return info
    except Exception as exc:  # the bench line must survive a collective problem
        return {'ok': False, 'error': repr(exc)[:300]}


def _watchdog(rdv_dir, n, t_start, limits):
    """Why the attempt must be stopped, or None.  The ranks write their milestones into rdv_dir/progress.<rank>:
    `ctx_ready` (library loaded, device context up) within limits['start'] of the launch; `gather_done` (communicator up
    and the sample all-gather checked -- the FIRST thing a multi-rank run does) within limits['gather'] of the last
    rank's ctx_ready; after that some rank must report a milestone at least every limits['stall'] seconds."""
    now = time.time()
    marks = []
    for r in range(n):
        try:
            with open(os.path.join(rdv_dir, 'progress.%d' % r)) as f:
                rows = []
                for ln in f.read().splitlines():        # the ranks append while we read: a half-written line is skipped
                    parts = ln.split(None, 1)
                    if len(parts) != 2:
                        continue
                    try:
                        rows.append((float(parts[0]), parts[1]))
                    except ValueError:
                        continue
                marks.append(rows)
        except OSError:
            marks.append([])
    ready = [next((t for t, m in mk if m == 'ctx_ready'), None) for mk in marks]
    if any(t is None for t in ready):
        return ('a rank did not bring its device context up within %.0f s' % limits['start']) if now - t_start > limits['start'] else None
    done = [any(m == 'gather_done' for _, m in mk) for mk in marks]
    if not all(done):
        if now - max(ready) > limits['gather']:
            return 'the communicator / sample all-gather check did not finish within %.0f s (ranks %s)' % (
                limits['gather'], [r for r in range(n) if not done[r]])
        return None
    last = max(mk[-1][0] for mk in marks)
    if now - last > limits['stall']:
        return 'no rank reported a milestone for %.0f s' % limits['stall']
    return None


def _run_ranks(args, argv, extra_env, limits):
    """One attempt: start the N ranks, wait.  Returns (rc, why it was stopped or None, rank-0 stdout lines, all ranks'
    lines for --spawn-check).  Rank 0's stdout is held back until the attempt is known to have succeeded."""
    import shutil
    import socket
    import tempfile
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    rdv_dir = tempfile.mkdtemp(prefix='mjhmc_bench_')
    procs, relays, collected = [], [], [[] for _ in range(n)]

    def relay(r, pipe):
        for line in iter(pipe.readline, b''):
            if args.spawn_check or r == 0:
                collected[r].append(line)
            else:
                sys.stderr.buffer.write(line)
                sys.stderr.buffer.flush()
        pipe.close()

    rc, timed_out = 0, None
    t_start = time.time()
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), MJHMC_COMM_ID_FILE=os.path.join(rdv_dir, 'comm.id'),
                       MJHMC_BENCH_PROGRESS=os.path.join(rdv_dir, 'progress'), MJHMC_BENCH_SPAWNED='1')
            env.update(extra_env)
            env.pop('TORCHELASTIC_USE_AGENT_STORE', None)
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=subprocess.PIPE)
            procs.append(p)
            t = threading.Thread(target=relay, args=(r, p.stdout), daemon=True)
            t.start()
            relays.append(t)
        alive = set(range(n))
        while alive and rc == 0:
            time.sleep(0.05)
            for r in sorted(alive):
                code = procs[r].poll()
                if code is not None:
                    alive.discard(r)
                    if code != 0:
                        rc = code if code > 0 else 1
                        sys.stderr.write('bench.py: rank %d exited with status %d; stopping the other ranks\n' % (r, code))
            if alive and limits:
                why = _watchdog(rdv_dir, n, t_start, limits)
                if why:
                    rc, timed_out = 124, why
                    sys.stderr.write('bench.py: %s; stopping the ranks\n' % why)
        for r in sorted(alive):                       # only after a failure: our own children, by pid
            procs[r].terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for t in relays:
            t.join(timeout=10)
    finally:
        for p in procs:                               # whatever ended the attempt (an exception here included): no rank is left behind
            if p.poll() is None:
                p.kill()
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pass
        shutil.rmtree(rdv_dir, ignore_errors=True)
    return rc, timed_out, collected


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with WORLD_SIZE unset: this process touches no GPU and does not load the library; it
    starts the N ranks as child processes (never exec), each with RANK / LOCAL_RANK / WORLD_SIZE, MASTER_ADDR / MASTER_PORT
    (only the gloo safety net reads them) and MJHMC_COMM_ID_FILE = a path in a fresh directory through which rank 0
    publishes the RCCL id; prints rank 0's stdout (the ONE JSON line), sends the other ranks' stdout to stderr, and exits
    non-zero if any rank does (ending the others: a rank that died would leave them waiting at the rendezvous).

    A scaling run must yield its line even where RCCL does not come up: when the first attempt fails, or its watchdog
    (_watchdog: device context within 240 s, communicator + sample all-gather check within 120 s, then a milestone at
    least every MJHMC_BENCH_RCCL_TIMEOUT = 300 s -- a hung collective cannot be cancelled from inside a rank) stops it,
    the ranks are started once more, as fresh processes, with the barrier / MAX and the sample gather on
    torch.distributed gloo (host-staged), and the line says so in config.comm_note.  The timed regions contain no
    collective either way."""
    if args.spawn_check:
        rc, _, collected = _run_ranks(args, argv, {}, None)
        if rc == 0:
            reports = [json.loads(b''.join(c).decode().strip().splitlines()[-1]) for c in collected]
            print(json.dumps({'spawn_check': reports, 'n_gpus': args.gpus, 'rendezvous_dir_fresh': True}))
        return rc
    forced = os.environ.get('MJHMC_BENCH_BACKEND')
    # the watchdog of the first (RCCL) attempt; DESIGN.md section 8 has the worst-case wall time these add up to
    limits = {'start': float(os.environ.get('MJHMC_BENCH_START_TIMEOUT', '240')),    # fresh box: library + first GPU touch
              'gather': float(os.environ.get('MJHMC_BENCH_GATHER_TIMEOUT', '120')),   # ncclCommInitRank + the gather check
              'stall': float(os.environ.get('MJHMC_BENCH_RCCL_TIMEOUT', '300'))}      # longest silence between milestones
    attempts = [({}, None)] if forced else [({}, limits), ({'MJHMC_BENCH_BACKEND': 'gloo'}, None)]
    rc = 1
    for k, (extra, lim) in enumerate(attempts):
        if k:
            extra = dict(extra, MJHMC_BENCH_NOTE='first attempt (RCCL) %s; this run: barrier / MAX / sample gather through '
                                                 'torch.distributed gloo' % why)
        rc, stopped, collected = _run_ranks(args, argv, extra, lim)     # (always fresh child processes: never re-exec a rank)
        if rc != 0 and not stopped and extra.get('MJHMC_BENCH_BACKEND') == 'gloo':
            # the gloo rendezvous port was picked by bind-then-close: another process may have taken it in between.  Once more, on a new one
            sys.stderr.write('bench.py: the gloo attempt failed with status %d; once more on a fresh port\n' % rc)
            rc, stopped, collected = _run_ranks(args, argv, extra, lim)
        if rc == 0:
            sys.stdout.buffer.write(b''.join(collected[0]))
            sys.stdout.buffer.flush()
            return 0
        why = ('was stopped: ' + stopped) if stopped else 'failed with status %d' % rc
        if k + 1 < len(attempts):
            sys.stderr.write('bench.py: attempt %d %s; starting the ranks again on the gloo safety net\n' % (k + 1, why))
    return rc


def spawn_report():
    """--spawn-check: what this rank was given, and whether the rendezvous channel works -- through the same
    FileRendezvous the RCCL id travels by -- without touching a GPU or loading the library."""
    from mjhmc_amd.parallel import FileRendezvous, default_id_path
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    path, explicit = default_id_path()
    if explicit:
        rdv = FileRendezvous(os.path.dirname(path), rank, world, timeout=60)
    else:
        rdv = StoreRendezvous(rank, world)
    if rank == 0:
        rdv.put('probe', b'id-from-rank-0')
    seen = rdv.get('probe').decode()
    agree = rdv.all_agree('probe_ok', seen == 'id-from-rank-0')
    print(json.dumps({'rank': rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '-1')), 'world': world,
                      'id_file': path, 'id_file_explicit': bool(explicit), 'probe': seen, 'all_ranks_agree': bool(agree),
                      'channel': type(rdv).__name__,
                      'library_loaded': any('libmjhmc_hip' in ln for ln in open('/proc/self/maps')),
                      'master': '%s:%s' % (os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT'))}))
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64)     # iterations per mjhmc_iterate call (one fused launch of the elementwise kernels)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--workload', default='all', choices=sorted(WORKLOADS) + ['all'])      # c3f64: C3 in the reference's arithmetic
    ap.add_argument('--head', default='c3f64', choices=['c2', 'c3', 'c3f64', 'c4', 'c5', 'c5bf16'],
                    help='top-level workload of the line (default: C3 in the reference\'s arithmetic -- float64 state around '
                         'the float32 force --, the workload of BASELINE.json\'s numeric target; c3: its float32-state form)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shard-of', default='auto',
                    help='comma list of G: after every workload, the same workload at N/G particles on this ONE GPU (what a rank '
                         'of a G-GPU strong-scaled run executes); shard_efficiency = t(N) / (G t(N/G)).  Default "auto": 2,4,8 '
                         'for the head, C4 and C5 (BASELINE.json words those as sharded; north_star: "fraction reported at '
                         '1/2/4/8"), 8 for the others; "1" = none.  Ignored at --gpus > 1')
    ap.add_argument('--detail', default=os.path.join(ROOT, 'gpurun_out', 'bench_detail.json'),
                    help='file that receives the complete per-workload records (the printed line carries numbers only)')
    ap.add_argument('--spawn-check', action='store_true',
                    help='ranks report their environment and the rendezvous channel, then exit before touching a GPU')
    args = ap.parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return spawn_ranks(args, argv)
    if args.spawn_check:
        return spawn_report()
    if os.environ.get('MJHMC_BENCH_FAKE'):
        return fake_rank(args)
    rig = Rig(args)
    progress('comm_up' if rig.world > 1 else 'single')
    # the one collective of the path FIRST: a broken or hanging all-gather is found in the run's first seconds (the
    # spawning parent watches for the marker below), not after five workloads
    gather_info = sample_gather_check(rig) if rig.comm is not None else None
    progress('gather_done')
    keys = ['c2', 'c3', 'c3f64', 'c4', 'c5', 'c5bf16'] if args.workload == 'all' else [args.workload]
    head = args.head if args.head in keys else keys[0]
    keys = [head] + [k for k in keys if k != head]
    rig.head = head
    results, strong, shards = {}, {}, {}
    def shard_gs(key):
        if rig.world != 1 or key == 'c1':
            return []
        if args.shard_of == 'auto':
            return [2, 4, 8] if key in (head, 'c4', 'c5') else [8]
        return sorted({int(g) for g in args.shard_of.split(',') if g.strip() and int(g) > 1})

    def budget(key):
        # the head workload runs exactly --steps / --warmup; the other dense workloads (10-20 ms per iteration: a batch
        # of K of them is already long) are capped
        if key == head or key in ('c2', 'c4', 'c1'):
            return args.steps, args.warmup
        return max(2, min(args.steps, 16)), min(args.warmup, 4)

    # ONE CPU baseline serves both forms of C3: the NumPy port integrates float64 state around a float32 force, which is
    # c3f64's arithmetic exactly; it is timed with whichever of the two runs as the head (else with c3f64)
    cpu_key = {'c3': head if head in ('c3', 'c3f64') else 'c3f64', 'c3f64': head if head in ('c3', 'c3f64') else 'c3f64',
               'c5': head if head in ('c5', 'c5bf16') else 'c5', 'c5bf16': head if head in ('c5', 'c5bf16') else 'c5'}
    # run order: the two vector-pipe / HBM workloads first, then the matrix-core ones -- whichever is the head.  (Measured:
    # C2 right after the ProductOfT run reads 7 % slower than on a chip that has not just run 20 s of dense MFMA work.)
    for key in [k for k in ('c1', 'c2', 'c4', 'c3', 'c3f64', 'c5', 'c5bf16') if k in keys]:
        steps, warm = budget(key)
        cpu_s = 0 if (args.no_cpu_baseline or cpu_key.get(key, key) != key) else (12.0 if key == head else 6.0)
        results[key] = run_workload(rig, key, steps, warm, cpu_s, args.scaling)
        progress('workload_done ' + key)
        if results[key] is not None:
            results[key]['steps'] = steps
            results[key]['warmup'] = warm
        # what a rank of a G-GPU strong-scaled run executes: the same workload at N/G particles on this one GPU, right
        # after the full-size run (same box, same clocks): shard_efficiency = t(N) / (G * t(N/G))
        for g in shard_gs(key):
            rec = run_workload(rig, key, steps, warm, 0, args.scaling, shard_of=g)
            progress('shard_done %s/%d' % (key, g))
            if rec is not None:
                rec.update(steps=steps, warmup=warm)
                rec['shard_efficiency'] = results[key]['ms_per_step'] / (g * rec['ms_per_step'])
                shards.setdefault(key, {})[g] = rec
    if rig.world > 1 and args.scaling == 'weak':
        # BASELINE.json words C4 and C5 as totals sharded over the GPUs: the same line carries them strong-scaled
        for key in [k for k in ('c4', 'c5') if k in keys]:
            steps, warm = budget(key)
            strong[key] = run_workload(rig, key, steps, warm, 0, 'strong')
            progress('strong_done ' + key)
            if strong[key] is not None:
                strong[key].update(steps=steps, warmup=warm, scaling='strong')
    if rig.rank == 0:
        # the two forms of C3 share the CPU baseline (which IS the float64-state arithmetic)
        for pair in ([results.get('c3'), results.get('c3f64')], [results.get('c5'), results.get('c5bf16')]):
            src = next((r for r in pair if r and 'cpu_baseline' in r), None)
            for r in pair:
                if r and src and 'cpu_baseline' not in r:
                    r['cpu_baseline'] = dict(src['cpu_baseline'], shared=True)
                    r['config']['gpu_over_cpu'] = r['value'] / src['cpu_baseline']['value']
        h = results[head]
        hr = h['roofline']
        # The printed line: the contract's keys, numbers only, <= 8 kB (the driver keeps the line's tail and the scalar keys
        # of `config`): every workload's ms / fraction / bound as FLAT scalars in `config`, compact per-workload records
        # under `workloads`, and the summary once more as the LAST key.  The complete records go to --detail.
        cfg = dict(h['config'])
        flat = {}
        for k, v in results.items():
            if not v:
                continue
            r, c = v['roofline'], v['config']
            flat['%s_ms' % k] = v['ms_per_step']
            flat['%s_frac' % k] = r['frac']                       # needed flops (bytes) / time / peak: one meaning everywhere
            flat['%s_bound' % k] = r['bound']
            if 'executed' in r:
                flat['%s_frac_executed' % k] = r['executed']['frac']
            if 'counted' in r:
                flat['%s_frac_counted' % k] = r['counted']['frac']
            if 'l2_stream' in r:
                flat['%s_l2stream_frac' % k] = r['l2_stream']['frac']
            if 'one_iteration_per_launch' in r:               # the sampling_iteration() path: one iteration per call (HBM-bound)
                flat['%s_one_iter_ms' % k] = r['one_iteration_per_launch']['avg_launch_ms']
                flat['%s_one_iter_frac' % k] = r['one_iteration_per_launch']['frac']
                flat['%s_one_iter_call_ms' % k] = r['one_iteration_per_launch'].get('call_ms')
            flat['%s_lfr' % k] = '%.3f/%.3f/%.3f' % (c['L_move_fraction'], c['F_move_fraction'], c['R_move_fraction'])
            if c.get('gpu_over_cpu'):
                flat['%s_gpu_over_cpu' % k] = c['gpu_over_cpu']
            if 'boundary' in v:                                   # particle-steps/s of sample(n): iterations + streamed download
                flat['%s_sample%d_incl_download' % (k, v['boundary']['samples'])] = v['boundary']['particle_steps_per_s_incl_download']
            if 'autocor_on_device' in v:
                flat['%s_autocor_on_device' % k] = v['autocor_on_device']['particle_steps_per_s_incl_autocor']
            for g, rec in sorted(shards.get(k, {}).items()):
                flat['%s_shard%d_ms' % (k, g)] = rec['ms_per_step']
                flat['%s_shard%d_frac' % (k, g)] = rec['roofline']['frac']
                flat['%s_shard%d_eff' % (k, g)] = rec['shard_efficiency']
        if 'c5' in results and results['c5']:
            flat['c5_law'] = 'kept (temperature within 1-3 %: bf16 matrix-core operands)'   # tests/test_gpu_stationary.py::test_sic_stationary_law
        if 'c5bf16' in results and results['c5bf16']:
            flat['c5bf16_law'] = 'hot'         # the bf16-state MarkovJumpHMC chain does not keep its law; c5 does
        cfg.update(flat)
        if gather_info is not None:
            cfg['sample_gather_ok'] = bool(gather_info.get('ok'))
            cfg['sample_gather_backend'] = gather_info.get('backend')
        if rig.comm_note:
            cfg['comm_note'] = rig.comm_note[:200]
        out = {
            'metric': 'particle-steps/sec (ndims x nparticles x L)',
            'value': h['value'], 'unit': h['unit'], 'n_gpus': rig.world, 'steps': h['steps'], 'warmup': h['warmup'],
            'ms_per_step': h['ms_per_step'], 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': h['dtype'], 'data': 'synthetic', 'config': cfg,
            'roofline': {k: hr[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'avg_launch_ms',
                                            'launches_timed', 'algorithmic_flops_per_launch', 'algorithmic_bytes_per_launch',
                                            'traffic_over_algorithmic', 'frac_at_held_clock') if k in hr},
            'timing': {k: h[k] for k in ('ms_per_step_median', 'ms_per_step_min', 'ms_per_step_max', 'repeats', 'timed_s')},
        }
        if 'executed' in hr:    # what the device ran / time (the line's `frac`: what the chain needs)
            out['roofline']['frac_executed'] = hr['executed']['frac']
        if 'counted' in hr:     # the reference's evaluation count / time (it integrates the F-movers' inverse-L proposals too)
            out['roofline']['frac_counted'] = hr['counted']['frac']
        if 'cpu_baseline' in h:
            b = h['cpu_baseline']
            out['cpu_baseline'] = {'value': b['value'], 'unit': b['unit'], 'cores': b['cores'], 'kind': b['kind'],
                                   'sample': 'NumPy port of the reference path, %d of %d columns, %d sampling_iterations after 1 '
                                             'warm-up, %.1f s, os.cpu_count=%d' % (b['nparticles'], WORKLOADS[head]['N'], b['iterations'],
                                                                                  b['seconds'], os.cpu_count())}
        tgt = results.get('c3f64') or results.get('c3')
        if tgt is not None and 'cpu_baseline' in tgt:
            # BASELINE.json: ">= 50x the NumPy reference in particle-steps/sec on ProductOfT (ndims=512, 100k particles) at 1 GPU"
            # -- quoted like for like: the GPU run in the reference's arithmetic against the NumPy port (the same arithmetic)
            ratio = tgt['value'] / tgt['cpu_baseline']['value']
            out['target'] = {'workload': 'c3f64' if results.get('c3f64') else 'c3', 'min_gpu_over_cpu': 50, 'gpu_over_cpu': ratio,
                             'met': bool(ratio >= 50)}
        if 'boundary' in h:
            out['boundary'] = h['boundary']
        ac = {k: v['autocor_on_device'] for k, v in results.items() if v and 'autocor_on_device' in v}
        if ac:
            out['autocor_on_device'] = ac
        if gather_info is not None:
            out['sample_gather'] = {k: v for k, v in gather_info.items() if k not in ('compared_with',)}
        if len(keys) > 1:
            out['workloads'] = {k: compact(v) for k, v in results.items() if v}
        if strong:
            out['strong'] = {k: compact(v) for k, v in strong.items() if v}
        # LAST key (the tail of the line is what a truncating reader keeps): [ms, frac, bound, shard-8 efficiency] per workload
        out['summary'] = {k: [v['ms_per_step'], v['roofline']['frac'], v['roofline']['bound'],
                              (shards.get(k, {}).get(8) or {}).get('shard_efficiency')] for k, v in results.items() if v}
        out['line_note'] = 'frac = needed flops (bytes) / time / peak; *_frac_executed, *_frac_counted beside it; c5 = float32 state, c5bf16_law = hot'

        line = json.dumps(rnd(out))
        print(line)
        sys.stdout.flush()
        if args.detail:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
                with open(args.detail, 'w') as f:
                    json.dump({'line_bytes': len(line), 'workloads': results,
                               'shards': {k: {str(g): r for g, r in v.items()} for k, v in shards.items()},
                               'strong': strong, 'sample_gather': gather_info}, f, indent=1, default=float)
            except OSError as exc:
                sys.stderr.write('bench.py: --detail %s not written (%s)\n' % (args.detail, exc))
    if rig.comm is not None:
        rig.comm.barrier()
        if hasattr(rig.comm, 'close'):
            rig.comm.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
