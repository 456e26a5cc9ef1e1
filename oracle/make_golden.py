#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (the reference does not travel):

    python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]

The reference (pure Python on NumPy) is loaded from where it lies under the
alias ``decomp_ref``; nothing of it is copied.  Its one missing dependency,
``chainer`` (imported but never used on the hot path: utils/cp_compat.py:1,
template_matching.py:2), is satisfied by empty in-process stub modules.
The fixtures hold DATA only: seeded inputs and the reference's outputs.
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np


def load_reference(ref_root):
    for name in ('chainer', 'chainer.cuda', 'chainer.utils',
                 'chainer.utils.conv', 'chainer.utils.conv_nd'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['chainer'].cuda = sys.modules['chainer.cuda']
    sys.modules['chainer'].utils = sys.modules['chainer.utils']
    sys.modules['chainer.utils'].conv = sys.modules['chainer.utils.conv']
    sys.modules['chainer.utils'].conv_nd = sys.modules['chainer.utils.conv_nd']
    pkg_dir = os.path.join(ref_root, 'decomp')
    spec = importlib.util.spec_from_file_location(
        'decomp_ref', os.path.join(pkg_dir, '__init__.py'),
        submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules['decomp_ref'] = mod
    spec.loader.exec_module(mod)
    import decomp_ref.nmf, decomp_ref.lasso, decomp_ref.dictionary_learning  # noqa
    return mod


# ------------------------------------------------------------------ NMF ----
def nmf_inputs(seed, N, F, K, dtype, likelihood):
    """Generator of tests/test_nmf.py:61-69 at an arbitrary shape."""
    rng = np.random.RandomState(seed)
    Dt = np.maximum(rng.randn(K, F), 0.0)
    xt = np.maximum(rng.randn(N, K), 0.0)
    y = xt.dot(Dt)
    noise = rng.randn(N, F) * 0.1
    y = y + (np.abs(noise) if likelihood == 'kl' else noise)
    D0 = np.maximum(Dt + rng.randn(K, F) * 0.3, 0.1)
    mask = np.rint(rng.uniform(0.3, 1, size=N * F)).reshape(N, F)
    return y.astype(dtype), D0.astype(dtype), mask.astype(dtype)


def gen_nmf(ref, out):
    from decomp_ref.nmf_methods import grads
    from decomp_ref.utils import normalize
    data = {}
    cases = []
    n_trace = 25
    for (N, F, K) in [(64, 48, 4), (101, 20, 3), (256, 128, 8)]:
        for dtype in (np.float64, np.float32):
            for lik in ('l2', 'kl'):
                for use_mask in (False, True):
                    seed = 0
                    y, D0, mask = nmf_inputs(seed, N, F, K, dtype, lik)
                    m = mask if use_mask else None
                    name = 'nmf_%dx%dk%d_%s_%s_%s' % (
                        N, F, K, np.dtype(dtype).name, lik,
                        'mask' if use_mask else 'nomask')
                    # per-iteration trace, driving the reference's own update
                    # rules exactly as batch_mu.py:16-24 does
                    likobj = grads.get_likelihood(lik)
                    D = normalize.l2_strict(D0, axis=-1, xp=np)
                    x = np.ones((N, K), dtype=dtype)
                    diffs, resids = [], []
                    for _ in range(n_trace):
                        x = likobj.update_x(y, x, D, m)
                        Dn = normalize.l2_strict(likobj.update_d(y, x, D, m),
                                                 axis=-1, xp=np)
                        diffs.append(np.max(np.abs(D - Dn)))
                        r = y - x.dot(Dn)
                        if m is not None:
                            r = r * m
                        resids.append(np.sqrt(np.sum(r.astype(np.float64) ** 2)))
                        D = Dn
                    # the full public solve
                    tol = 1.0e-6 if dtype == np.float64 else 1.0e-4
                    it, Dfin, xfin = ref.nmf.solve(
                        y, D0.copy(), x=None, tol=tol, minibatch=None,
                        maxiter=400, method='mu', likelihood=lik, mask=m,
                        random_seed=0)
                    cases.append(name)
                    base = name.rsplit('_', 1)[0]     # inputs stored once
                    data[base + '/y'] = y
                    data[base + '/D0'] = D0
                    data[base + '/mask'] = mask
                    data[name + '/trace_maxdiff'] = np.array(diffs, np.float64)
                    data[name + '/trace_resid'] = np.array(resids, np.float64)
                    data[name + '/trace_D'] = D
                    data[name + '/trace_x'] = x
                    data[name + '/tol'] = np.float64(tol)
                    data[name + '/it'] = np.int64(it)
                    data[name + '/D'] = Dfin
                    data[name + '/x'] = xfin
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(out, 'nmf_golden.npz'), **data)
    print('nmf: %d cases' % len(cases))


def gen_nmf_minibatch(ref, out):
    """Stochastic variants (tests/test_nmf.py:105-152 shapes: 1001 x 20, K = 3,
    minibatch 30), a few epochs with tol = 0 so the whole trajectory is pinned."""
    data = {}
    cases = []
    methods = ['asg-mu', 'gsg-mu', 'asag-mu', 'gsag-mu', 'svrmu', 'svrmu-acc']
    for dtype in (np.float64, np.float32):
        for lik in ('l2', 'kl'):
            if dtype == np.float32 and lik == 'kl':
                continue
            y, D0, mask = nmf_inputs(0, 1001, 20, 3, dtype, lik)
            base = 'nmfmb_%s_%s' % (np.dtype(dtype).name, lik)
            data[base + '/y'] = y
            data[base + '/D0'] = D0
            data[base + '/mask'] = mask
            for method in methods:
                for use_mask in (False, True):
                    for maxiter in (3,):
                        it, D, x = ref.nmf.solve(
                            y.copy(), D0.copy(), x=None, tol=0.0, minibatch=30,
                            maxiter=maxiter, method=method, likelihood=lik,
                            mask=mask.copy() if use_mask else None, random_seed=0)
                        name = '%s/%s/%s/it%d' % (base, method, 'mask' if use_mask else 'nomask',
                                                  maxiter)
                        cases.append(name)
                        data[name + '/it'] = np.int64(it)
                        data[name + '/D'] = D
                        data[name + '/x'] = x
            # one converging run per method (the old-D-on-convergence quirk)
            for method in methods:
                it, D, x = ref.nmf.solve(y.copy(), D0.copy(), x=None, tol=3.0e-2, minibatch=30,
                                         maxiter=30, method=method, likelihood=lik, mask=None,
                                         random_seed=0)
                name = '%s/%s/conv' % (base, method)
                cases.append(name)
                data[name + '/it'] = np.int64(it)
                data[name + '/D'] = D
                data[name + '/x'] = x
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(out, 'nmf_minibatch_golden.npz'), **data)
    print('nmf minibatch: %d cases' % len(cases))


# ---------------------------------------------------------------- LASSO ----
def lasso_inputs(seed, batch_shape, K, F, kind):
    """Generators of tests/test_lasso.py:160-250 (vector / matrix / tensor;
    real / complex / float32)."""
    rng = np.random.RandomState(seed)

    def randn(*s):
        if kind == 'c128':
            return rng.randn(*s) + rng.randn(*s) * 1.0j
        return rng.randn(*s)

    A = randn(K, F)
    n = int(np.prod(batch_shape)) if batch_shape else 1
    xt = randn(n * K) * np.rint(rng.uniform(size=n * K))
    xt = xt.reshape(tuple(batch_shape) + (K,))
    y = np.dot(xt, A) + randn(*(tuple(batch_shape) + (F,))) * 0.1
    mask = np.rint(rng.uniform(0.4, 1, size=n * F)).reshape(
        tuple(batch_shape) + (F,))
    mask1d = np.rint(rng.uniform(0.4, 1.0, size=F))
    if kind == 'f32':
        A, y = A.astype(np.float32), y.astype(np.float32)
        mask, mask1d = mask.astype(np.float32), mask1d.astype(np.float32)
    return y, A, mask, mask1d


def gen_lasso(ref, out):
    data = {}
    cases = []
    real_methods = ['ista', 'acc_ista', 'fista', 'cd',
                    'ista_pos', 'acc_ista_pos', 'fista_pos', 'cd_pos']
    cplx_methods = ['ista', 'acc_ista', 'fista', 'cd']
    shapes = {'vec': (), 'mat': (11,), 'ten': (12, 11)}
    for kind in ('f64', 'c128', 'f32'):
        methods = cplx_methods if kind == 'c128' else real_methods
        for sname, bshape in shapes.items():
            y, A, mask, mask1d = lasso_inputs(0, bshape, 5, 10, kind)
            base = 'lasso_%s_%s' % (kind, sname)
            data[base + '/y'] = y
            data[base + '/A'] = A
            data[base + '/mask2d'] = mask
            data[base + '/mask1d'] = mask1d
            for mname, m in (('nomask', None), ('mask1d', mask1d),
                             ('mask2d', mask)):
                for method in methods:
                    for (tol, maxiter, tag) in ((1.0e-6, 1000, 'conv'),
                                                (1.0e-9, 7, 'exh')):
                        if kind == 'f32' and tag == 'conv':
                            tol = 1.0e-5
                        alpha = 0.1
                        it, x = ref.lasso.solve(
                            y.copy(), A.copy(), alpha=alpha, tol=tol,
                            method=method, maxiter=maxiter,
                            mask=None if m is None else m.copy())
                        name = '%s/%s/%s/%s' % (base, mname, method, tag)
                        cases.append(name)
                        data[name + '/it'] = np.int64(it)
                        data[name + '/x'] = np.asarray(x)
                        data[name + '/tol'] = np.float64(tol)
                        data[name + '/maxiter'] = np.int64(maxiter)
                        data[name + '/alpha'] = np.float64(alpha)
    # prox known-answer vectors of tests/test_lasso.py:15-56
    z = np.array([[0.1, -2.0, 1.4], [1.1, 3.0, -1.4]])
    data['prox/z'] = z
    data['prox/real'] = ref.lasso.soft_threshold_float(z, 1.0, np)
    data['prox/complex45'] = ref.lasso.soft_threshold_complex(
        z + z * 1.0j, 1.0, np)
    data['prox/positive'] = ref.lasso.soft_threshold_positive(z, 1.0, np)
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(out, 'lasso_golden.npz'), **data)
    print('lasso: %d cases' % len(cases))


def lasso_inputs_wide(seed, N, K, F, kind, correlated=False):
    """A wider problem than the reference's 5 x 10 test design, so that parallel_cd commits
    several coordinates per iteration; `correlated` makes the atoms nearly parallel
    (Gershgorin bound ~ K, p <= 1: the fallback branch of lasso.py:469-470)."""
    rng = np.random.RandomState(seed)

    def randn(*s):
        if kind in ('c128', 'c64'):
            return rng.randn(*s) + rng.randn(*s) * 1.0j
        return rng.randn(*s)

    A = randn(K, F)
    if correlated:
        A = randn(1, F) + 0.05 * A
    xt = randn(N, K) * np.rint(rng.uniform(size=(N, K)))
    y = np.dot(xt, A) + randn(N, F) * 0.1
    mask = np.rint(rng.uniform(0.4, 1, size=(N, F)))
    if kind == 'f32':
        A, y, mask = A.astype(np.float32), y.astype(np.float32), mask.astype(np.float32)
    if kind == 'c64':
        A, y, mask = A.astype(np.complex64), y.astype(np.complex64), mask.astype(np.float32)
    return y, A, mask


def gen_lasso_extra(ref, out):
    """parallel_cd and admm: lasso.py:448-523, 586-657."""
    data = {}
    cases = []

    def run(name, y, A, m, method, tol, maxiter, alpha=0.1):
        try:
            it, x = ref.lasso.solve(y.copy(), A.copy(), alpha=alpha, tol=tol, method=method,
                                    maxiter=maxiter, mask=None if m is None else m.copy())
            data[name + '/it'] = np.int64(it)
            data[name + '/x'] = np.asarray(x)
            data[name + '/raises'] = np.array('')
        except TypeError as e:      # lasso.py:509 (masked parallel_cd fallback)
            data[name + '/it'] = np.int64(-1)
            data[name + '/x'] = np.zeros(0)
            data[name + '/raises'] = np.array('TypeError')
        cases.append(name)
        data[name + '/tol'] = np.float64(tol)
        data[name + '/maxiter'] = np.int64(maxiter)
        data[name + '/alpha'] = np.float64(alpha)

    shapes = {'vec': (), 'mat': (11,), 'ten': (12, 11)}
    for kind in ('f64', 'c128', 'f32'):
        methods = ['parallel_cd', 'admm']
        if kind != 'c128':
            methods += ['parallel_cd_pos', 'admm_pos']
        for sname, bshape in shapes.items():
            y, A, mask, mask1d = lasso_inputs(0, bshape, 5, 10, kind)
            base = 'lasso_%s_%s' % (kind, sname)       # same inputs as lasso_golden.npz
            data[base + '/y'] = y
            data[base + '/A'] = A
            data[base + '/mask2d'] = mask
            data[base + '/mask1d'] = mask1d
            for mname, m in (('nomask', None), ('mask1d', mask1d), ('mask2d', mask)):
                if m is not None and m.ndim > 1 and y.ndim == 1:
                    pass
                for method in methods:
                    for (tol, maxiter, tag) in ((1.0e-6, 1000, 'conv'), (1.0e-9, 7, 'exh')):
                        if kind == 'f32' and tag == 'conv':
                            tol = 1.0e-5
                        run('%s/%s/%s/%s' % (base, mname, method, tag), y, A, m, method, tol,
                            maxiter)
    for kind in ('f64', 'f32', 'c128', 'c64'):
        for tag, corr in (('wide', False), ('corr', True)):
            y, A, mask = lasso_inputs_wide(3, 16, 24, 64, kind, correlated=corr)
            base = 'lasso_%s_%s' % (kind, tag)
            data[base + '/y'] = y
            data[base + '/A'] = A
            data[base + '/mask2d'] = mask
            methods = ['parallel_cd', 'admm'] if tag == 'wide' else ['parallel_cd']
            if kind[0] == 'f':
                methods = methods + [m + '_pos' for m in methods]
            for mname, m in (('nomask', None), ('mask2d', mask)):
                for method in methods:
                    tol = 1.0e-5 if kind in ('f32', 'c64') else 1.0e-7
                    run('%s/%s/%s/conv' % (base, mname, method), y, A, m, method, tol, 400)
                    run('%s/%s/%s/exh' % (base, mname, method), y, A, m, method, 1.0e-12, 13)
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(out, 'lasso_extra_golden.npz'), **data)
    print('lasso extra: %d cases' % len(cases))


# ------------------------------------------------- dictionary learning -----
def dl_inputs(seed, complex_):
    """Generator of tests/test_dictionary.py:35-43."""
    rng = np.random.RandomState(seed)

    def randn(*s):
        if complex_:
            return rng.randn(*s) + rng.randn(*s) * 1.0j
        return rng.randn(*s)

    Dt = randn(3, 5)
    xt = randn(101, 3)
    xt = xt * rng.uniform(size=303).reshape(101, 3)
    y = np.dot(xt, Dt) + randn(101, 5) * 0.1
    D0 = Dt + randn(3, 5) * 0.2
    mask = np.rint(rng.uniform(0.45, 1, size=505)).reshape(101, 5)
    return y, D0, mask


def gen_dl(ref, out):
    data = {}
    cases = []
    for complex_ in (False, True):
        y, D0, mask = dl_inputs(0, complex_)
        base = 'dl_%s' % ('c128' if complex_ else 'f64')
        data[base + '/y'] = y
        data[base + '/D0'] = D0
        data[base + '/mask'] = mask
        for minibatch in (100, 10):
            for lasso_method, lasso_iter in (('ista', 10), ('acc_ista', 30),
                                             ('fista', 10), ('cd', 10)):
                for use_mask in (False, True):
                    if use_mask and (minibatch != 10 or lasso_method == 'cd'):
                        continue
                    for epochs in (1, 2, 3):
                        yy = y * mask if use_mask else y
                        it, D, x = ref.dictionary_learning.solve(
                            yy.copy(), D0.copy(), 0.1, x=None, tol=0.0,
                            minibatch=minibatch, maxiter=epochs + 1,
                            lasso_method=lasso_method, lasso_iter=lasso_iter,
                            lasso_tol=1.0e-5, random_seed=0,
                            mask=mask.copy() if use_mask else None)
                        name = '%s/mb%d/%s%d/%s/ep%d' % (
                            base, minibatch, lasso_method, lasso_iter,
                            'mask' if use_mask else 'nomask', epochs)
                        cases.append(name)
                        data[name + '/it'] = np.int64(it)
                        data[name + '/D'] = D
                        data[name + '/x'] = x
        # the reference test's own configuration (tests/test_dictionary.py:45-56)
        it, D, x = ref.dictionary_learning.solve(
            y.copy(), D0.copy(), 0.1, x=None, tol=1.0e-4, minibatch=100,
            maxiter=1000, lasso_method='acc_ista', lasso_iter=1000,
            random_seed=0)
        name = base + '/reftest'
        cases.append(name)
        data[name + '/it'] = np.int64(it)
        data[name + '/D'] = D
        data[name + '/x'] = x
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(out, 'dl_golden.npz'), **data)
    print('dictionary learning: %d cases' % len(cases))


def dl_inputs_wide(seed, N, F, K, complex_):
    """The recipe of tests/test_dictionary.py:35-43 at a wider dictionary (K > 64 atoms: several
    blocks of the blocked atom sweep, csrc/atom_sweep.hpp)."""
    rng = np.random.RandomState(seed)

    def randn(*s):
        if complex_:
            return rng.randn(*s) + rng.randn(*s) * 1.0j
        return rng.randn(*s)

    Dt = randn(K, F)
    xt = randn(N, K) * (rng.uniform(size=N * K).reshape(N, K) < 0.05)
    y = np.dot(xt, Dt) + randn(N, F) * 0.1
    D0 = Dt + randn(K, F) * 0.2
    return y, D0


def gen_dl_extra(ref, out):
    """(1) float32 / complex64 runs of the dl_golden cases (BASELINE configs[2] and [4] dtypes);
    (2) wide dictionaries (K = 160 real: two full 64-atom blocks + a 32-atom tail; K = 80 complex),
    all dtypes, so that the blocked atom sweep is pinned to the real reference."""
    data = {}
    cases = []
    for complex_ in (False, True):
        y64, D064, mask64 = dl_inputs(0, complex_)
        cdt = np.complex64 if complex_ else np.float32
        y, D0, mask = y64.astype(cdt), D064.astype(cdt), mask64.astype(np.float32)
        base = 'dl_%s' % ('c64' if complex_ else 'f32')
        data[base + '/y'] = y
        data[base + '/D0'] = D0
        data[base + '/mask'] = mask
        for minibatch in (100, 10):
            for lasso_method, lasso_iter in (('ista', 10), ('acc_ista', 30), ('cd', 10)):
                for use_mask in (False, True):
                    if use_mask and (minibatch != 10 or lasso_method == 'cd'):
                        continue
                    for epochs in (1, 2):
                        yy = y * mask if use_mask else y
                        it, D, x = ref.dictionary_learning.solve(
                            yy.copy(), D0.copy(), 0.1, x=None, tol=0.0,
                            minibatch=minibatch, maxiter=epochs + 1,
                            lasso_method=lasso_method, lasso_iter=lasso_iter,
                            lasso_tol=1.0e-5, random_seed=0,
                            mask=mask.copy() if use_mask else None)
                        name = '%s/mb%d/%s%d/%s/ep%d' % (
                            base, minibatch, lasso_method, lasso_iter,
                            'mask' if use_mask else 'nomask', epochs)
                        cases.append(name)
                        data[name + '/it'] = np.int64(it)
                        data[name + '/D'] = D
                        data[name + '/x'] = x
    for tag, complex_, cdt, K in (('f64', False, np.float64, 160), ('f32', False, np.float32, 160),
                                  ('c128', True, np.complex128, 80), ('c64', True, np.complex64, 80)):
        N, F = 384, 48
        y64, D064 = dl_inputs_wide(1, N, F, K, complex_)
        y, D0 = y64.astype(cdt), D064.astype(cdt)
        base = 'dlwide_%s' % tag
        data[base + '/y'] = y
        data[base + '/D0'] = D0
        for lasso_method, lasso_iter in (('ista', 10), ('cd', 10)):
            for epochs in (1, 2):
                it, D, x = ref.dictionary_learning.solve(
                    y.copy(), D0.copy(), 0.02, x=None, tol=0.0, minibatch=128,
                    maxiter=epochs + 1, lasso_method=lasso_method,
                    lasso_iter=lasso_iter, lasso_tol=1.0e-5, random_seed=0)
                name = '%s/%s%d/ep%d' % (base, lasso_method, lasso_iter, epochs)
                cases.append(name)
                data[name + '/it'] = np.int64(it)
                data[name + '/D'] = D
                data[name + '/x'] = x
    data['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(out, 'dl_extra_golden.npz'), **data)
    print('dictionary learning (f32 / c64 / wide): %d cases' % len(cases))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(
        os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.ref, 'decomp')):
        print('reference not present at %s: nothing to do' % args.ref)
        return 0
    os.makedirs(args.out, exist_ok=True)
    ref = load_reference(args.ref)
    gen_nmf(ref, args.out)
    gen_nmf_minibatch(ref, args.out)
    gen_lasso(ref, args.out)
    gen_lasso_extra(ref, args.out)
    gen_dl(ref, args.out)
    gen_dl_extra(ref, args.out)
    return 0


if __name__ == '__main__':
    sys.exit(main())
