"""The engine's multi-PROCESS path on one GPU.

RCCL refuses two ranks on one device, and the builder's boxes have one GPU: the sharded kernels run in an in-process loopback group
(test_gpu_sharded.py), but the code a real job goes through - one process per rank, the unique-id rendezvous, gnn_comm_create, the RCCL
call sites (grouped all-gathers of state rows / boundary blocks and flags, the all-to-alls of the feature-sliced exchange, the
point-to-point schedule of its pipelined return, the max all-reduce) - had never executed with more than one rank.  Here it does, over a
stand-in transport (tests/mock_rccl: the same entry points, data through /dev/shm, blocking), selected with GNN_RCCL_LIBRARY.  What this
checks: ORDER, peers, counts and offsets of every call on every rank (a mismatch deadlocks into the transport's time-out or gives wrong
rows) and the results against the C oracle.  What it cannot check: RCCL itself, xGMI, overlap, timing.  Also: `bench.py --gpus N`
end to end through its own launcher with all ranks on device 0 (GNN_BENCH_ONE_DEVICE=1)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import c_oracle as corc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, 'tests', 'mock_rccl', 'libmock_rccl.so')


def _mock():
    if not os.path.exists(MOCK):
        subprocess.run(['make', '-C', os.path.dirname(MOCK)], check=True, capture_output=True)
    return MOCK


def _env(rank, world, extra=None):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), GNN_RCCL_LIBRARY=_mock(), OMP_NUM_THREADS='2')
    env.update(extra or {})
    return env


@pytest.mark.parametrize('world', [2, 4])
def test_ranks_as_processes_match_oracle(world):
    """world processes, every exchange layout, impl 1 bit-identical to the C oracle (k, states, outputs), impl 2 within tolerance; then the
    training step on shards - forward (state all-gather per body, BatchNormalization statistics and gates of all ranks) and backward (the
    aggregate-column gradients all-gathered per body, BatchNormalization sums of all ranks, the ranks' weight-gradient shares added in rank
    order) - against the float64 oracle and the one-GPU forward."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_gpu_sharded as S
    n, d = 4099, 8
    g, st, ou, s0 = S._case(4242, n, d, hidden=(16,))
    kc, sc, oc = corc.loop_node(g, st, ou, d, 30, 0.01, s0)
    with tempfile.TemporaryDirectory() as out:
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_mp_worker.py'), out], env=_env(r, world), cwd=ROOT,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        logs = []
        for p in procs:
            try:
                logs.append(p.communicate(timeout=420)[0])
            except subprocess.TimeoutExpired:
                for q in procs: q.kill()
                raise
        assert all(p.returncode == 0 for p in procs), '\n'.join(l[-1500:] for l in logs)
        res = [np.load(os.path.join(out, f'rank{r}.npz')) for r in range(world)]
    layouts = ['full', 'halo'] + (['slice1', 'slice2'] if d % world == 0 else [])
    for layout in layouts:
        for impl in (1, 2):
            ks = {float(r[f'{layout}_{impl}_k']) for r in res}
            assert ks == {float(kc)}, (layout, impl, ks, kc)
            state = np.concatenate([r[f'{layout}_{impl}_state'] for r in res])
            outp = np.concatenate([r[f'{layout}_{impl}_out'] for r in res])
            assert state.shape == sc.shape and outp.shape == oc.shape
            if impl == 1:
                assert np.array_equal(state, sc) and np.array_equal(outp, oc), (layout, int(np.sum(state != sc)))
            else:
                assert np.max(np.abs(state - sc)) < 2e-6 * max(1.0, float(np.max(np.abs(sc)))) and np.max(np.abs(outp - oc)) < 2e-6, layout
            assert {float(r[f'{layout}_{impl}_max']) for r in res} == {float(world)}      # gnn_comm_allreduce_max over the ranks


    # ---- graph readout on shards: every rank returns the same [G, T], the oracle's up to the rounding of graphs that straddle a shard boundary
    gr_, stg, oug, s0g = S._case(911, 960, 8)
    gr_['set_mask'] = np.ones(960, bool); gr_['output_mask'] = np.ones(960, bool)
    rngg = np.random.default_rng(12)
    bounds = np.sort(rngg.choice(np.arange(1, 960), 11, replace=False))
    sizes = np.diff(np.concatenate([[0], bounds, [960]]))
    ng_indptr = np.concatenate([[0], np.cumsum(sizes)])
    kg, sg, og = corc.loop_node(gr_, stg, oug, 8, 20, 0.01, s0g)
    node_graph = np.zeros((960, 12), np.float32)
    for gi in range(12): node_graph[ng_indptr[gi]:ng_indptr[gi + 1], gi] = 1.0 / sizes[gi]
    want = corc.readout(node_graph, og)
    assert {float(r['readout_k']) for r in res} == {float(kg)}
    assert all(np.array_equal(res[0]['readout'], r['readout']) for r in res[1:])
    assert res[0]['readout'].shape == want.shape and np.max(np.abs(res[0]['readout'] - want)) < 1e-6

    # ---- the two-layer LGNN stack with the relabelling between the layers across the rank processes, against the C oracle chain
    from oracle import gnn_oracle as orc
    from util import make_mlp as _mk
    rng1 = np.random.default_rng(31)
    nl1 = g['nodes'].shape[1] + d + 2
    st1 = _mk(rng1, 1 + 2 * (d + nl1), [16, d], 'selu', gain=0.6, bn_random=True)
    ou1 = _mk(rng1, d + nl1, [2], 'softmax', bn_random=True)
    s1 = (0.1 * rng1.standard_normal((n, d))).astype(np.float32)
    k0c, s0c, o0c = corc.loop_node(g, st, ou, d, 20, 0.01, s0)
    g1 = orc.update_graph(g, s0c, o0c, True, True)
    k1c, s1c, o1c = corc.loop_node(g1, st1, ou1, d, 20, 0.01, s1)
    for layout in ('full', 'halo'):
        assert {float(r[f'lgnn_{layout}_k0']) for r in res} == {float(k0c)} and {float(r[f'lgnn_{layout}_k1']) for r in res} == {float(k1c)}, layout
        assert np.array_equal(np.concatenate([r[f'lgnn_{layout}_labels'] for r in res]), g1['nodes']), layout
        assert np.array_equal(np.concatenate([r[f'lgnn_{layout}_state'] for r in res]), s1c), layout
        assert np.array_equal(np.concatenate([r[f'lgnn_{layout}_out'] for r in res]), o1c), layout

    # ---- the training-mode forward on shards against the float64 oracle of the WHOLE graph (BatchNormalization over all rows) and
    # against the same forward on one GPU
    from oracle import gnn_train_oracle as tro
    from util import make_mlp
    from GNN import _engine as e
    for tag, (nt, dt, hidden, thr_t) in (('train', (1531, 8, (16,), 0.02)), ('trainw', (12000, 64, (128, 128), 0.0))):
        gt, stt, out_, s0t = S._case(777 + nt, nt, dt, hidden=hidden)
        rngt = np.random.default_rng(nt)
        stt = make_mlp(rngt, stt['weights'][0].shape[0], list(hidden) + [dt], 'selu' if tag == 'train' else 'tanh', gain=0.6, bn_random=True)      # (wide case: a smooth activation - SELU's kink makes single gradient entries jump)
        out_ = make_mlp(rngt, out_['weights'][0].shape[0], [2], 'softmax', bn_random=True)
        stt['dropout'], out_['dropout'] = {}, {}
        ctx = tro.train_forward(gt, stt, out_, dt, 4, thr_t, s0t, [{}] * 4, {})
        ks = {float(r[f'{tag}_k']) for r in res}
        assert ks == {float(ctx['k'])} and ctx['k'] >= 2, (tag, ks, ctx['k'])
        state = np.concatenate([r[f'{tag}_state'] for r in res])
        outp = np.concatenate([r[f'{tag}_out'] for r in res])
        scale = max(1.0, float(np.max(np.abs(ctx['state']))))
        assert state.shape == ctx['state'].shape and outp.shape == ctx['out_nodes'].shape
        assert np.max(np.abs(state - ctx['state'])) < 2e-5 * scale and np.max(np.abs(outp - ctx['out_nodes'])) < 2e-5, \
            (tag, float(np.max(np.abs(state - ctx['state']))), float(np.max(np.abs(outp - ctx['out_nodes']))))
        # one GPU: the order in which the chunk statistics of BatchNormalization are merged differs, and with few rows per rank the dense
        # products run on the FP32 ALUs instead of the matrix cores (gnn_train.hip, tg_many_rows): float32-level differences
        ipt, srct, wt, awt, alt = S._csr_parts(gt)
        maskt = np.logical_and(gt['set_mask'], gt['output_mask'])
        mst, mou = e.Mlp(stt['weights'], stt['activations'], True), e.Mlp(out_['weights'], out_['activations'], True)
        lp = e.Loop(e.Graph(nt, ipt, srct, wt, awt, alt, gt['nodes'], maskt), mst, mou, dt, 4, thr_t)
        lp.set_state0(s0t)
        k1, out1 = lp.train_forward(mst, mou, None, bn_state=np.concatenate(stt['weights'][-4:-2]), bn_output=np.concatenate(out_['weights'][-4:-2]))
        so_ = max(1.0, float(np.max(np.abs(out1))))
        assert k1 == ctx['k'] and np.max(np.abs(lp.state() - state)) < 2e-5 * scale and np.max(np.abs(out1 - outp)) < 2e-5 * so_
        # ---- the backward half on the shards: every rank returns the same, complete gradients; against the float64 oracle's step
        m_all = int(maskt.sum())
        rngl = np.random.default_rng(99)
        targets = np.eye(2)[rngl.integers(0, 2, m_all)].astype(np.float32)
        weights = (rngl.uniform(0.5, 1.5, m_all) / m_all).astype(np.float32)
        ref = tro.train_step(gt, stt, out_, dt, 4, thr_t, s0t, [{}] * 4, {}, targets, weights, loss='categorical_crossentropy', mean=False, graph_based=False)
        assert abs(sum(float(r[f'{tag}_loss']) for r in res) - ref['loss']) < 1e-5 * max(1.0, abs(ref['loss']))
        gscale = max(float(np.max(np.abs(w_))) for w_ in ref['grads_state'])
        for name, wl in (('gs', ref['grads_state']), ('go', ref['grads_output'])):
            for i, want in enumerate(wl):
                got = [r[f'{tag}_{name}{i}'] for r in res]
                assert all(np.array_equal(got[0], x) for x in got[1:]), (tag, name, i)          # the same bits on every rank
                tol = 1e-3 * max(float(np.max(np.abs(want))), 0.1 * gscale)
                assert got[0].shape == want.shape and np.max(np.abs(got[0] - want)) < tol, (tag, name, i, float(np.max(np.abs(got[0] - want))), tol)


@pytest.mark.parametrize('gpus,exchange', [(2, 'auto'), (4, 'auto'), (4, 'slice1'), (3, 'halo')])
def test_bench_launcher_runs_all_ranks_end_to_end(gpus, exchange):
    """`python bench.py --gpus N` bare: the launcher starts N rank processes, they rendezvous, shard the graph, run the timed Loops through
    the communicator and rank 0 prints the JSON line (a reduced graph; all ranks on device 0 over the stand-in transport)."""
    env = dict(os.environ, GNN_RCCL_LIBRARY=_mock(), GNN_BENCH_ONE_DEVICE='1', OMP_NUM_THREADS='2')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'): env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--steps', '2', '--warmup', '1', '--nodes', '60000',
           '--exchange', exchange, '--no-cpu-baseline', '--no-other-configs']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == gpus and line['steps'] == 2 and line['scaling'] == 'strong' and line['value'] > 0
    # the N > 1 line validates itself: 3 bodies sharded vs unsharded on every rank (bit for bit on the exact path), and it says what an
    # iteration is made of
    mg = line['multi_gpu']
    assert line['parity_ok'] is True and mg['sharded_check']['ok'] is True
    assert mg['sharded_check']['sharded_vs_unsharded_max_abs_diff'] == 0.0 and mg['sharded_check']['k_equal'] is True
    assert mg['exchange'] in ('full', 'slice', 'slice1', 'halo') and mg['kernel_ms_per_iteration'] > 0 and mg['exchange_ms_per_iteration'] > 0
    if mg['exchange'] == 'full':
        assert mg['bytes_received_per_rank_per_iteration'] == (gpus - 1) * (line_nodes(line) // gpus) * 64 * 4


def line_nodes(line):
    import re
    return int(re.search(r'N=(\d+) nodes', line['config']['workload']).group(1))


def test_bench_under_torch_distributed_run():
    """The driver's own invocation for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the agent; the RCCL id through a file keyed by the
    agent's pid and the port).  Two ranks on device 0 over the stand-in transport, a reduced graph."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, GNN_RCCL_LIBRARY=_mock(), GNN_BENCH_ONE_DEVICE='1', OMP_NUM_THREADS='2')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'GNN_BENCH_RDV'): env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--nodes', '60000', '--no-cpu-baseline', '--no-other-configs']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1 and line['scaling'] == 'strong' and line['value'] > 0
