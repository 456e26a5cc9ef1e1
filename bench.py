#!/usr/bin/env python3
"""Benchmark of the headline metric (BASELINE.json): NMF multiplicative-update
iterations/s at Y = 65536 x 4096, k = 256, float32, on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is ONE MU iteration (x update, D update, l2_strict, max|dD| stop test) of the
whole job; rows of Y / x are sharded over the ranks (strong scaling: the problem size is
fixed) and the [K, F+K] statistics are all-reduced once per step (RCCL).  Inputs are
synthetic (SURVEY 8d recipe) and resident in HBM before the timed region.

Rank 0 prints one JSON line.  `roofline` is the dominant kernel (the fused
Y.D^T + quotient GEMM) timed with HIP events on the solver's own stream inside the timed
steps; `cpu_baseline` is the NumPy oracle (the reference's formulation) on a bounded
sample of the same workload on this host's cores (N = 1, rank 0 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS, N_FEAT, N_ATOMS = 65536, 4096, 256
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 MFMA peak
PEAK_HBM_GBS = 8000.0


def synth(n_rows, rank_seed, device):
    """Y = xt.Dt + 0.1|noise| (>= 0), D0 = max(Dt + 0.3 noise, 0.1): SURVEY 8d, C2.
    Generated on the GPU (data synthesis only; not part of the measured path)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(1234)                       # Dt / D0 identical on every rank
    Dt = torch.randn((N_ATOMS, N_FEAT), generator=g, device=device).clamp_(min=0)
    D0 = (Dt + 0.3 * torch.randn((N_ATOMS, N_FEAT), generator=g, device=device)).clamp_(min=0.1)
    g.manual_seed(99 + rank_seed)             # this rank's rows
    xt = torch.randn((n_rows, N_ATOMS), generator=g, device=device).clamp_(min=0)
    Y = xt @ Dt
    Y += 0.1 * torch.randn((n_rows, N_FEAT), generator=g, device=device).abs_()
    del xt
    return Y, D0


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    summary (profiles/*_pmc_summary.json: (c*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc
    passes; c = the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md, 2.0 for wide
    coalesced reads and 1.10 as calibrated for the K-contiguous panel loader's 64-byte row
    segments, profiles/r01_fetch_calibration.txt).  PMC counters cannot be read inside this
    process, so the number is the one measured by tools/profile_round.sh on this same
    command at N = 1; None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_summary.json')))
    if not files:
        return None, 'no committed PMC summary'
    try:
        d = json.load(open(files[-1]))
        for k, v in d.items():
            if kernel_key in k and 'hbm_bytes_corrected' in v:
                return v['hbm_bytes_corrected'], 'from %s (N=1 run)' % os.path.basename(files[-1])
    except Exception as e:  # pragma: no cover
        return None, 'unreadable PMC summary: %s' % e
    return None, 'kernel not in PMC summary'


def cpu_baseline(budget_s=20.0):
    """The oracle (NumPy restatement of the reference's 6-GEMM formulation) on a bounded
    row sample of the same workload, scaled to whole-job iterations/s."""
    import numpy as np
    from oracle import nmf as onmf, common
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    cores, blas = affinity, 'unknown'
    try:    # the threads the BLAS behind NumPy really runs (what `cores` must state)
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get('user_api') == 'blas']
        if pools:
            cores = int(max(p.get('num_threads', 1) for p in pools))
            blas = '%s %s' % (pools[0].get('internal_api'), pools[0].get('version'))
    except Exception:
        pass
    rows = 4096
    rng = np.random.RandomState(0)
    Dt = np.maximum(rng.randn(N_ATOMS, N_FEAT), 0).astype(np.float32)
    xt = np.maximum(rng.randn(rows, N_ATOMS), 0).astype(np.float32)
    y = xt @ Dt + 0.1 * np.abs(rng.randn(rows, N_FEAT)).astype(np.float32)
    D = common.l2_strict(np.maximum(Dt + 0.3 * rng.randn(N_ATOMS, N_FEAT), 0.1).astype(np.float32))
    x = np.ones((rows, N_ATOMS), np.float32)
    x, D, _ = onmf.mu_step(y, x, D)           # warm-up (BLAS threads, page faults)
    t0 = time.perf_counter()
    iters = 0
    while iters < 3 or (time.perf_counter() - t0 < budget_s and iters < 400):
        x, D, _ = onmf.mu_step(y, x, D)
        iters += 1
    per_iter = (time.perf_counter() - t0) / iters
    scale = N_ROWS / rows
    return {'value': 1.0 / (per_iter * scale), 'unit': 'iterations/s', 'cores': cores,
            'kind': 'port', 'blas': blas, 'affinity_cores': affinity,
            'sample': 'oracle.nmf.mu_step (NumPy/BLAS, reference 6-GEMM formulation), %d of '
                      '%d rows x %d iterations (%.0f s), time scaled x%d'
                      % (rows, N_ROWS, iters, per_iter * iters, scale)}


def secondary_configs(torch, device):
    """The other BASELINE configs at their one-GPU shapes, measured live in a few seconds each (they
    are parity-test cases, not the headline; reported so that their numbers in DESIGN.md have a
    driver-side record).  Synthetic data of SURVEY 8d's recipes."""
    from decomp_amd import _arrays, _hip
    out = {}
    g = torch.Generator(device=device)
    g.manual_seed(2)

    def ms_of(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # configs[2]: one dictionary-learning minibatch step, 8192 x 4096, k = 512, ista x 10, fp32
    MB, F, K = 8192, 4096, 512
    Dt = torch.randn((K, F), generator=g, device=device)
    xt = 30.0 * torch.randn((MB, K), generator=g, device=device) * \
        (torch.rand((MB, K), generator=g, device=device) < 0.05)
    Y = xt @ Dt + 0.1 * torch.randn((MB, F), generator=g, device=device)
    D = Dt + 0.2 * torch.randn((K, F), generator=g, device=device)
    _arrays.l2_normalize_(D, strict=True)
    x = torch.ones((MB, K), device=device)
    A = torch.zeros((K, K), device=device)
    B = torch.zeros((K, F), device=device)
    D_new = torch.empty_like(D)
    lib, h = _arrays.lib_handle(D)
    md, lit = ctypes.c_double(0), ctypes.c_int(0)
    state = {'D': D, 'Dn': D_new, 'count': 0}

    def dl_step():
        theta = state['count'] * MB + 1.0
        _hip.check(h, lib.dcp_dict_step_f32(h, _arrays.ptr(Y), _arrays.ptr(x), _arrays.ptr(state['D']),
                                            _arrays.ptr(state['Dn']), _arrays.ptr(A), _arrays.ptr(B), MB, F, K,
                                            (theta - MB) / theta, 0.1, _hip.LASSO_ISTA, 10, 1e-5,
                                            ctypes.byref(md), ctypes.byref(lit)), 'dict_step')
        state['D'], state['Dn'] = state['Dn'], state['D']
        state['count'] += 1
    out['dictionary_step_ms'] = {'workload': 'configs[2] minibatch 8192x4096 k=512 ista x10 fp32',
                                 'value': round(ms_of(dl_step, 6), 4)}
    del Y, x, A, B, D, D_new, Dt, xt

    # configs[3]: masked NMF MU, one GPU's shard 16384 x 4096, k = 256, 20 % missing, fp32
    N, F, K = 16384, 4096, 256
    Dt = torch.randn((K, F), generator=g, device=device).clamp_(min=0)
    xt = torch.randn((N, K), generator=g, device=device).clamp_(min=0)
    Y = xt @ Dt + 0.1 * torch.randn((N, F), generator=g, device=device).abs_()
    D = (Dt + 0.3 * torch.randn((K, F), generator=g, device=device)).clamp_(min=0.1)
    mask = (torch.rand((N, F), generator=g, device=device) >= 0.2).float()
    _arrays.l2_normalize_(D, strict=True)
    x = torch.ones((N, K), device=device)
    it = ctypes.c_int(0)

    def masked(n=5):
        _hip.check(h, lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), _arrays.ptr(mask), _arrays.ptr(x), _arrays.ptr(D),
                                         N, F, K, _hip.LIK_L2, ctypes.c_float(0.0), n + 1, ctypes.byref(it),
                                         None, None), 'nmf_mu masked')
    ms = ms_of(masked, 2) / 5
    out['masked_nmf_ms_per_iter'] = {'workload': 'configs[3] shard 16384x4096 k=256 20% mask fp32',
                                     'value': round(ms, 4),
                                     'tflops_on_12NKF': round(12.0 * N * K * F / ms / 1e9, 1)}
    # float64 (the reference's default dtype) on the fp64 MFMA core, same shard shape, no mask
    Yd, Dd = Y.double(), D.double()
    xd = torch.ones((N, K), device=device, dtype=torch.float64)

    def f64(n=5):
        _hip.check(h, lib.dcp_nmf_mu_f64(h, _arrays.ptr(Yd), None, _arrays.ptr(xd), _arrays.ptr(Dd), N, F, K,
                                         _hip.LIK_L2, ctypes.c_double(0.0), n + 1, ctypes.byref(it), None,
                                         None), 'nmf_mu f64')
    ms = ms_of(f64, 2) / 5
    W = 4.0 * N * K * F + 4.0 * N * K * K + 4.0 * K * K * F
    out['f64_nmf_ms_per_iter'] = {'workload': '16384x4096 k=256 float64', 'value': round(ms, 4),
                                  'tflops': round(W / ms / 1e9, 1), 'fp64_peak': 78.6}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the few-second side measurements of the other BASELINE configs')
    ap.add_argument('--no-kernel-events', action='store_true',
                    help='(analysis only) do not bracket kernel groups with hipEvents in the timed steps')
    ap.add_argument('--force-sharded', action='store_true',
                    help='(analysis only) use the multi-GPU Python step loop even at N = 1')
    ap.add_argument('--rows', type=int, default=0,
                    help='(analysis only) rows of Y on this GPU instead of 65536/N: time one '
                         'shard of a larger run without the collective')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from decomp_amd import _arrays, _hip, sharded

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
        args.gpus = world
    # one rank per GPU; DCP_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the
    # launch path on a 1-GPU box -- never a measurement)
    backend = os.environ.get('DCP_DIST_BACKEND', 'nccl')
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    rows = N_ROWS // world
    if args.rows:
        rows = args.rows
    Y, D0 = synth(rows, rank, device)
    x = torch.ones((rows, N_ATOMS), dtype=torch.float32, device=device)
    state = {'x': x}
    D = D0.clone()
    _arrays.l2_normalize_(D, strict=True)
    torch.cuda.synchronize()

    lib, h = _arrays.lib_handle(D)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def run(n_steps):
        """n_steps MU iterations (tol = 0: the stop test is evaluated, never met)."""
        if world == 1 and not args.force_sharded:
            it = ctypes.c_int(0)
            rc = lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(D),
                                    rows, N_FEAT, N_ATOMS, _hip.LIK_L2, ctypes.c_float(0.0),
                                    n_steps + 1, ctypes.byref(it), None, None)
            _hip.check(h, rc, 'dcp_nmf_mu_f32')
            assert it.value == n_steps + 1
            return D
        backend = sharded.HipStepBackend(Y, None, state['x'], D, _hip.LIK_L2)
        it, Dout = sharded.mu_loop(backend, D, 0.0, n_steps + 1, world_size=world,
                                   new_like=torch.empty_like)
        assert it == n_steps + 1
        state['x'] = backend.x
        return Dout

    def read_profile():
        out = {}
        for lab in range(_hip.PROF_NLABELS):
            ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            _hip.check(h, lib.dcp_profile_read(h, lab, ctypes.byref(ms), ctypes.byref(cnt)), 'profile_read')
            if cnt.value:
                out[lib.dcp_profile_label_name(lab).decode()] = {
                    'ms_total': ms.value, 'launches': cnt.value, 'ms_avg': ms.value / cnt.value}
        return out

    Dcur = run(args.warmup)
    if Dcur is not D:
        D.copy_(Dcur)
    # Timed region: K steps; only the DOMINANT kernel is bracketed by hipEvents inside it (a
    # bracket costs ~4 us of stream time; bracketing all eight kernel groups costs 2.5 % of a
    # step at N = 1 and 12 % on an 8192-row shard -- measured with --no-kernel-events).
    _hip.check(h, lib.dcp_profile_reset(h), 'profile_reset')
    _hip.check(h, lib.dcp_profile_select(h, 1 << _hip.PROF_XUPDATE), 'profile_select')
    _hip.check(h, lib.dcp_profile_enable(h, 0 if args.no_kernel_events else 1), 'profile_enable')
    barrier()
    t0 = time.perf_counter()
    Dcur = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    _hip.check(h, lib.dcp_profile_enable(h, 0), 'profile_enable')

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    prof = read_profile()

    # After the timed region: a short untimed pass with every kernel group bracketed, for the
    # per-kernel breakdown (kernel_ms_avg); it does not enter `value`.
    if Dcur is not D:
        D.copy_(Dcur)
    _hip.check(h, lib.dcp_profile_reset(h), 'profile_reset')
    _hip.check(h, lib.dcp_profile_select(h, 0xffffffff), 'profile_select')
    _hip.check(h, lib.dcp_profile_enable(h, 1), 'profile_enable')
    Dcur = run(min(10, args.steps))
    barrier()
    _hip.check(h, lib.dcp_profile_enable(h, 0), 'profile_enable')
    breakdown = read_profile()
    prof_stats = breakdown.get('stats')

    finite = bool(torch.isfinite(Dcur).all().item()) and bool(torch.isfinite(state['x']).all().item())

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        W = 4.0 * N_ROWS * N_ATOMS * N_FEAT + 4.0 * N_ROWS * N_ATOMS ** 2 + 4.0 * N_ATOMS ** 2 * N_FEAT
        out = {
            'metric': 'nmf_mu_iterations_per_s', 'value': args.steps / elapsed,
            'unit': 'iterations/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('nmf_mu l2 no-mask Y=65536x4096 k=256 fp32 (BASELINE configs[1])'
                                    if not args.rows else 'ANALYSIS ONLY: one shard of %d rows' % rows),
                       'rows_per_gpu': rows, 'parallelism': 'rows sharded x%d, 1 all-reduce/step' % world},
            'algorithmic_tflops': W / (elapsed / args.steps) / 1e12,
            'mfma_roofline_frac_whole_step': W / (elapsed / args.steps) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
            'finite': finite,
            'kernel_ms_avg': {k: round(v['ms_avg'], 4) for k, v in breakdown.items()},
        }
        # dominant kernel: the fused Y.D^T GEMM + MU quotient (2.N.K.F flop per launch)
        dom = prof.get('x_update')
        traffic, traffic_src = pmc_traffic('0, 0, false, EpiMuNum<float>')
        if dom:
            flops = 2.0 * rows * N_ATOMS * N_FEAT
            ach = flops / (dom['ms_avg'] * 1e-3) / 1e12
            out['roofline'] = {'bound': 'mfma', 'kernel': 'gemm_mfma_kernel<NT, EpiMuNum> (Y.D^T + quotient)',
                               'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                               'traffic_note': traffic_src,
                               'algorithmic_bytes': 4.0 * (rows * N_FEAT + N_ATOMS * N_FEAT + 3 * rows * N_ATOMS),
                               'launch_ms': dom['ms_avg'], 'launches': dom['launches']}
        st = prof_stats
        if st:
            flops = 2.0 * rows * N_ATOMS * (N_FEAT + N_ATOMS)
            ach = flops / (st['ms_avg'] * 1e-3) / 1e12
            out['roofline_stats_gemm'] = {'bound': 'mfma', 'kernel': 'gemm_mfma_kernel<TN, EpiSlab> (x^T.[Y|x])',
                                          'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS,
                                          'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
                                          'launch_ms': st['ms_avg']}
        if world == 1 and not args.rows and not args.no_secondary:
            try:
                del Y
                torch.cuda.empty_cache()
                out['secondary'] = secondary_configs(torch, device)
            except Exception as e:      # never lose the headline line to a side measurement
                out['secondary'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
