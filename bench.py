#!/usr/bin/env python3
"""Benchmark of the GNN.Loop hot path on MI355X (BASELINE.json metric: node-state-updates/s).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N ...            (bare: this process touches no GPU, starts N rank processes itself and relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2]): synthetic graph with 1,000,000 nodes and ~10,000,000 directed arcs
(randomGraph recipe, 'average' aggregation), state_dim 64, node/arc labels 3/1, net_state 135->128->128->64 selu +
BatchNormalization, net_output 67->2 softmax + BatchNormalization, injected N(0, 0.1^2) initial state, threshold 0 so
that every Loop runs exactly max_iteration = 30 iterations.  One "step" = one GNN.Loop (condition, 30 x convergence,
apply_filters, net_output).  Graph, weights and initial state are resident in HBM before the timed region.

With N > 1 the SAME graph is sharded by node range (strong scaling): each rank owns N/P destination rows, and every
iteration ends with one grouped RCCL all-gather of the owned state rows + convergence flag.  No torch in the process: ranks
read RANK / LOCAL_RANK / WORLD_SIZE from the environment (set by either launcher) and exchange the RCCL id through a file.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


class Watchdog:
    """A rank that hangs inside a collective (a peer died, a wrong count or peer on some rank) must not hang the job: the ctypes calls
    release the GIL, so a timer thread can end THIS process (exit code 124); the launcher - bench.py's own or torch.distributed.run -
    then stops the other ranks and exits non-zero.  Fresh processes only, nothing is re-executed."""

    def __init__(self, seconds, what, rank):
        import threading
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.what, self.rank, self.seconds = what, rank, seconds

    def _fire(self):
        print(f'bench.py: rank {self.rank}: "{self.what}" did not finish within {self.seconds:.0f} s - giving up (exit 124)', file=sys.stderr, flush=True)
        os._exit(124)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def make_net(rng, n_in, widths, act, out_act=None, gain=1.0):
    """Random-init weights of the architecture (lecun_normal-like scale, as starter.py:52-54), BatchNormalization defaults.
    gain < 1 (other_configs only) makes the state map a contraction, so that Loops converge before max_iteration."""
    w, acts = [], []
    for i, u in enumerate(widths):
        w += [(gain * rng.standard_normal((n_in, u)) / np.sqrt(n_in)).astype(np.float32), (rng.standard_normal(u) / np.sqrt(u)).astype(np.float32) * 0.1]
        acts.append(out_act if (out_act and i == len(widths) - 1) else act)
        n_in = u
    w += [np.ones(n_in, np.float32), np.zeros(n_in, np.float32), np.zeros(n_in, np.float32), np.ones(n_in, np.float32)]
    return dict(weights=w, activations=acts, batch_normalization=True)


def algorithmic_bytes_per_iteration(n, e, ds, nl, al):
    """fp32 values, int32 indices, no cache credit for neighbour rows, fused iteration (no concat / aggregate
    materialised): E(4 Ds + 4 + 4) + 4(N + 1) + N(4 Ds read + 4 Ds write + 4(2 NL + AL) invariant label terms)."""
    return e * (4 * ds + 8) + 4 * (n + 1) + n * (8 * ds + 4 * (2 * nl + al))


def rendezvous_id(rank, world, engine):
    """Rank 0 creates the RCCL unique id; the others read it from a file keyed by the launcher process (same parent)."""
    if os.environ.get('GNN_BENCH_RDV'):                   # bare `bench.py --gpus N`: the launcher names the file
        return _rendezvous_file(os.environ['GNN_BENCH_RDV'], rank, engine)
    ppid = os.getppid()
    try:
        with open(f'/proc/{ppid}/stat') as f:
            start = f.read().rsplit(')', 1)[1].split()[19]
    except Exception:
        start = '0'
    return _rendezvous_file(f'/tmp/gnn_rccl_{ppid}_{start}_{os.environ.get("MASTER_PORT", "0")}_{world}.id', rank, engine)


def _rendezvous_file(path, rank, engine):
    if rank == 0:
        uid = engine.Comm.unique_id()
        with open(path + '.tmp', 'wb') as f:
            f.write(uid)
        os.replace(path + '.tmp', path)
        return uid, path
    deadline = time.time() + 300
    while not os.path.exists(path):
        if time.time() > deadline:
            raise RuntimeError(f'rank {rank}: timed out waiting for {path}')
        time.sleep(0.05)
    with open(path, 'rb') as f:
        return f.read(), path


def oracle_graph(s):
    n = s['n_nodes']
    arcs = np.concatenate([np.stack([s['src'], s['dst']], 1).astype(np.float32), s['arc_labels']], axis=1)
    return dict(nodes=s['nodes'], arcs=arcs, set_mask=np.ones(n, bool), output_mask=np.ones(n, bool),
                adjT=(s['indptr'], s['adj_src'], s['adj_w']), arcT=(s['indptr'], s['arc_perm'], s['arc_w']))


def cpu_baseline(s, st, ou, state0, d, iters, engine=None, device=0, full=None):
    """The plain-C restatement of the TF2 op sequence (oracle/gnn_oracle.c: CSR SpMM -> materialised concat -> Dense x3 ->
    BatchNormalization -> norm check), OpenMP over all host cores, on a bounded sample: the full graph, `iters` iterations;
    a second figure with the sparse products on ONE thread (TensorFlow's CPU SparseTensorDenseMatMul is believed to be
    single-threaded); and, as the checker it is, the float64 oracle against both GPU arithmetic modes at the depth
    of the workload on a 50,000-node graph of the same generator (fp32-noise equivalence of the default path, measured)."""
    from oracle import c_oracle, gnn_oracle
    n = s['n_nodes']
    g = oracle_graph(s)
    c_oracle.loop_node(g, st, ou, d, 1, 0.0, state0)     # warm-up (page faults, thread pool)
    t = time.perf_counter()
    k, s_orc, o_orc = c_oracle.loop_node(g, st, ou, d, iters, 0.0, state0)
    dt = time.perf_counter() - t
    c_oracle.set_spmm_single_thread(True)
    t1 = time.perf_counter()
    k1, _, _ = c_oracle.loop_node(g, st, ou, d, max(1, iters // 4), 0.0, state0)
    dt1 = time.perf_counter() - t1
    c_oracle.set_spmm_single_thread(False)
    out = dict(value=n * k / dt, unit='node-state-updates/s', cores=c_oracle.num_threads(), kind='port',
               sample=f'full 1M-node graph, {int(k)} iterations + readout, C restatement of the TF2 op sequence '
                      f'(not TensorFlow itself), OpenMP, {dt:.1f} s',
               single_thread_spmm={'value': n * k1 / dt1, 'sample': f'{int(k1)} iterations, sparse products on 1 thread, dense layers on '
                                                                     f'{c_oracle.num_threads()}, {dt1:.1f} s'})
    if full is not None:
        # the oracle is the checker: the GPU Loop on the SAME graph / weights / state0 for the same number of bodies, compared with
        # the oracle state that was just timed - all N x 64 values (impl 1: bit for bit; default path: max |difference|) - and with
        # the float64 shadow of the same Loop on the whole graph (oracle/gnn_oracle_f64.c, the arbiter of fp32 orders)
        graph, mst, mou = full
        t64 = time.perf_counter()
        k64, s64, o64 = c_oracle.loop_node_f64(g, st, ou, d, iters, 0.0, state0)
        dt64 = time.perf_counter() - t64
        chk = {'iterations': int(k), 'oracle_max_abs_state': float(np.max(np.abs(s_orc))),
               'float64_shadow': f'same Loop in double on the whole graph, {dt64:.1f} s; fp32 oracle vs float64: state '
                                 f'{float(np.max(np.abs(s_orc - s64))):.3e}, output {float(np.max(np.abs(o_orc - o64))):.3e}'}
        for impl, name in ((1, 'exact_f32_mfma_path'), (2, 'default_split_bf16_path')):
            lp = engine.Loop(graph, mst, mou, d, int(iters), 0.0)
            lp.set_impl(impl)
            lp.set_state0(state0)
            k_gpu = lp.run()
            sg, og_ = lp.state(), lp.output()
            lp.close()
            chk[name] = {'k_equal': bool(k_gpu == k), 'bit_identical_state': bool(np.array_equal(sg, s_orc)),
                         'max_abs_diff_state': float(np.max(np.abs(sg - s_orc))), 'max_abs_diff_output': float(np.max(np.abs(og_ - o_orc))),
                         'max_abs_state_error_vs_float64': float(np.max(np.abs(sg - s64))),
                         'max_abs_output_error_vs_float64': float(np.max(np.abs(og_ - o64)))}
            del sg, og_
        del s64, o64
        out['gpu_vs_oracle_full_size'] = chk
        # The verdict main() acts on, at the workload's own size and depth (the oracle ran `iters` bodies; the default is the workload's
        # max_iter).  Exact path: the oracle's bits (state, output, k).  Default path: the same k AND values that add nothing to the float32
        # noise - no further from float64 than 1.5 x the exact fp32 chain is, state and output (tests/test_gpu_full_size.py asserts the
        # same).  north_star's literal "1e-5 of the fp32 oracle" is recorded beside it: after 30 bodies of an expansive map (random-init
        # weights) no float32 ORDER meets it against another - the oracle's own chain is further than that from float64.
        ex, df = chk['exact_f32_mfma_path'], chk['default_split_bf16_path']
        noise_ok = bool(df['max_abs_state_error_vs_float64'] <= 1.5 * ex['max_abs_state_error_vs_float64']
                        and df['max_abs_output_error_vs_float64'] <= 1.5 * ex['max_abs_output_error_vs_float64'] + 1e-7)
        scale = max(1.0, chk['oracle_max_abs_state'])
        out['default_path_values'] = {
            'criterion': '|default - float64| <= 1.5 x |exact fp32 chain - float64|, state and output, whole graph, workload depth',
            'ok': noise_ok,
            'state_error_vs_float64': {'default': df['max_abs_state_error_vs_float64'], 'exact': ex['max_abs_state_error_vs_float64']},
            'output_error_vs_float64': {'default': df['max_abs_output_error_vs_float64'], 'exact': ex['max_abs_output_error_vs_float64']},
            'north_star_1e-5_vs_fp32_oracle': {'exact': bool(ex['bit_identical_state']),
                                               'default': bool(df['max_abs_diff_state'] <= 1e-5 * scale and df['max_abs_diff_output'] <= 1e-5),
                                               'default_max_abs': df['max_abs_diff_state'], 'scale_max_abs_state': scale}}
        out['parity_ok'] = bool(ex['k_equal'] and ex['bit_identical_state'] and ex['max_abs_diff_output'] == 0.0 and df['k_equal'] and noise_ok)
    del s_orc, o_orc
    if engine is not None:
        from GNN import GNN_utils as utils
        n2, bodies = 50_000, 30
        s2 = utils.syntheticGraph(n2, 10.0, 3, 1, 2, seed=20261003)
        rng = np.random.default_rng(7)
        s02 = (0.1 * rng.standard_normal((n2, d))).astype(np.float32)
        k64, s64, o64 = gnn_oracle.loop_node(oracle_graph(s2), st, ou, d, bodies, 0.0, s02, np.float64)
        graph = engine.Graph(n2, s2['indptr'], s2['adj_src'], s2['adj_w'], s2['arc_w'], s2['arc_labels_csr'], s2['nodes'], np.ones(n2, np.uint8), device=device)
        mst = engine.Mlp(st['weights'], st['activations'], True, device=device)
        mou = engine.Mlp(ou['weights'], ou['activations'], True, device=device)
        dist = {}
        for impl, name in ((1, 'exact_fp32_chain'), (2, 'default_split_bf16')):
            lp = engine.Loop(graph, mst, mou, d, bodies, 0.0)
            lp.set_impl(impl)
            lp.set_state0(s02)
            lp.run()
            dist[name + '_max_abs_state_error_vs_float64'] = float(np.max(np.abs(lp.state() - s64)))
            dist[name + '_max_abs_output_error_vs_float64'] = float(np.max(np.abs(lp.output() - o64)))
            lp.close()
        dist['graph'] = f'{n2} nodes / {s2["n_arcs"]} arcs, same generator and weights, {bodies} bodies (threshold 0), max |state| {float(np.max(np.abs(s64))):.2f}'
        out['fp32_noise_check'] = dist
    return out


def launch_ranks(n, argv, worker=None, grace_s=10.0):
    """Parent of a bare `python bench.py --gpus N` (N > 1).  It makes NO GPU call (the engine is not even imported here): it
    starts N fresh rank processes - RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, the RCCL
    id exchanged through the file named by GNN_BENCH_RDV - waits for them, relays rank 0's JSON line on stdout and returns 0;
    if any rank fails it stops the others (by their exact pids) and returns that rank's exit code.  `worker`: command prefix of a
    rank (default: this interpreter on this file); tests pass a stub."""
    worker = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix='gnn_bench_')
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GNN_BENCH_RDV=os.path.join(tmp, 'rccl.id'))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs, outs = [], []
    for r in range(n):
        out = open(os.path.join(tmp, f'rank{r}.out'), 'w+')
        outs.append(out)
        procs.append(subprocess.Popen(worker + list(argv), env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=out))      # stderr: inherited
    rc, live = 0, set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code
                print(f'bench.py: rank {r} of {n} exited with code {code}; stopping the other ranks', file=sys.stderr, flush=True)
                break
        time.sleep(0.05)
    if rc != 0:
        for r in live:
            procs[r].terminate()
        deadline = time.time() + grace_s
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    outs[0].seek(0)
    text0 = outs[0].read()
    for r, out in enumerate(outs):
        if r:                                   # ranks > 0 print nothing on stdout normally; whatever they did goes to stderr
            out.seek(0)
            extra = out.read()
            if extra.strip(): print(f'[rank {r} stdout] {extra}', file=sys.stderr)
        out.close()
    for name in os.listdir(tmp):
        os.remove(os.path.join(tmp, name))
    os.rmdir(tmp)
    if rc == 0:
        lines = [ln for ln in text0.splitlines() if ln.startswith('{')]
        if not lines:
            print(f'bench.py: rank 0 printed no JSON line; its stdout was: {text0[-2000:]!r}', file=sys.stderr)
            return 1
        for ln in text0.splitlines():
            if not ln.startswith('{') and ln.strip(): print(ln, file=sys.stderr)
        print(lines[-1], flush=True)
    elif text0.strip():
        print(text0, file=sys.stderr)
    return rc


def other_configs(engine, s, device=0):
    """The other BASELINE configs on this GPU, after the timed region (never part of `value`): configs[1] MUTAG batches of 32 graphs
    (gnn_loop_run - the persistent one-launch loop - plus the NodeGraph readout, as GNNgraphBased.Loop does), and configs[4] the
    5-layer LGNN stack on the 1M-node graph (get_state=False, get_output=True: labels widened by 2; relabelling on the device)."""
    from GNN.graph_class import GraphObject, GraphTensor
    import load_MUTAG
    out = {}
    rng = np.random.default_rng(1)
    # ---- configs[1]: MUTAG, net_state 31 -> 32 -> 32 -> 14 (D = 0), net_output 14 -> 2, max_iter 50, threshold 0.01 --------------
    graphs = load_MUTAG.load(limit=320)
    batches = [GraphTensor.fromGraphObject(GraphObject.merge(graphs[i:i + 32], problem_based='g', aggregation_mode='average')) for i in range(0, 320, 32)]
    def net(n_in, widths, act, gain):          # the weights tools/bench_small.py has always used (tests/util.make_mlp), so that rounds stay comparable
        w, d_ = [], n_in
        for u in widths:
            w += [(gain * rng.standard_normal((d_, u)) / np.sqrt(d_)).astype(np.float32), (rng.standard_normal(u) / np.sqrt(d_)).astype(np.float32)]
            d_ = u
        return dict(weights=w + [np.ones(d_, np.float32), np.zeros(d_, np.float32), np.zeros(d_, np.float32), np.ones(d_, np.float32)],
                    activations=[act] * len(widths), batch_normalization=True)
    st, ou = net(31, [32, 32, 14], 'selu', 0.7), net(14, [2], 'softmax', 1.0)
    mst, mou = engine.Mlp(st['weights'], st['activations'], True, device=device), engine.Mlp(ou['weights'], ou['activations'], True, device=device)
    loops = []
    for b in batches:
        lp = engine.Loop(b.device_graph(device), mst, mou, 0, 50, 0.01)
        loops.append((lp, b.nodegraph_csr(), b.nodes.shape[0]))
    persistent = all(lp.set_persistent(True) for lp, _, _ in loops)
    for lp, ng, _ in loops:
        lp.run(); lp.readout(*ng)
    reps, iters, updates = 30, 0.0, 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        for lp, ng, nn in loops:
            k = lp.run()
            lp.readout(*ng)
            iters += k; updates += k * nn
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()                            # the Loop alone (gnn_loop_run; what rounds 1 - 2 quoted as Loops/s)
    for _ in range(reps):
        for lp, _, _ in loops:
            lp.run()
    dt_loop = time.perf_counter() - t1
    t2 = time.perf_counter()                            # all ten batches in one call (gnn_loop_run_many: their launches side by side), then the readouts
    for _ in range(reps):
        engine.Loop.run_many([lp for lp, _, _ in loops])
        for lp, ng, _ in loops: lp.readout(*ng)
    dt_many = time.perf_counter() - t2
    out['mutag_batch32'] = {'loops_per_s': reps * len(loops) / dt, 'loop_only_per_s': reps * len(loops) / dt_loop,
                            'side_by_side_loops_per_s': reps * len(loops) / dt_many,
                            'graphs_per_s': 32 * reps * len(loops) / dt, 'node_state_updates_per_s': updates / dt,
                            'mean_iterations': iters / (reps * len(loops)), 'us_per_iteration': 1e6 * dt / iters, 'persistent_one_launch_loop': bool(persistent),
                            'what': 'BASELINE configs[1] shape: 10 batches of 32 MUTAG graphs (~570 nodes each), net_state 31->32->32->14, max_iter 50, '
                                    'threshold 0.01, GNN.Loop + NodeGraph readout per batch, host-timed (loop_only_per_s: without the readout; side_by_side_loops_per_s: the ten batches through gnn_loop_run_many + readouts)'}
    for lp, _, _ in loops: lp.close()
    # ---- configs[4]: LGNN x 5 on the bench graph ----------------------------------------------------------------------------------
    layers, d, nl, al, t, max_it = 5, 64, 3, 1, 2, 30
    n = s['n_nodes']
    base = engine.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8), device=device)
    derived = base.derive(t)
    stack = []
    rng = np.random.default_rng(7)
    for layer in range(layers):
        nll = nl + (t if layer else 0)
        st, ou = make_net(rng, al + 2 * (nll + d), [128, 128, d], 'selu'), make_net(rng, nll + d, [t], 'softmax')
        lp = engine.Loop(base if layer == 0 else derived, engine.Mlp(st['weights'], st['activations'], True, device=device),
                         engine.Mlp(ou['weights'], ou['activations'], True, device=device), d, max_it, 0.0)
        lp.set_state0(None, seed=layer + 1)              # drawn on the device: N(0, 0.1^2)
        stack.append(lp)

    def run_stack():
        ks = 0.0
        for layer, lp in enumerate(stack):
            ks += lp.run()
            if layer < layers - 1:
                derived.update_labels(base, lp, False, True)
        return ks

    run_stack()
    reps = 3
    t0 = time.perf_counter()
    ks = sum(run_stack() for _ in range(reps))
    dt = time.perf_counter() - t0
    out['lgnn_x5_1M'] = {'ms_per_lgnn_loop': 1e3 * dt / reps, 'iterations_per_lgnn_loop': ks / reps, 'node_state_updates_per_s': n * ks / dt,
                         'what': 'BASELINE configs[4]: 5 stacked GNN.Loops (30 bodies each, threshold 0) on the 1M-node graph, labels of layers 1-4 '
                                 'widened by the previous output (139->128->128->64 / 69->2), relabelling on the device, host-timed'}
    for lp in stack: lp.close()
    derived.close(); base.close()
    return out


def profile_figures():
    """Counter-derived figures of the dominant kernel from the committed rocprofv3 PMC passes of this command
    (tools/profile.sh + tools/collect_profile.py -> profiles/<round>_pmc.json); None when no profile is committed."""
    for tag in ('r05', 'r04', 'r03', 'r02', 'r01'):
        path = os.path.join(ROOT, 'profiles', f'{tag}_pmc.json')
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            d['source'] = f'profiles/{tag}_pmc.json'
            return d
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--nodes', type=int, default=1_000_000)
    ap.add_argument('--arcs-per-node', type=float, default=10.0)
    ap.add_argument('--max-iter', type=int, default=30)
    ap.add_argument('--impl', type=int, default=2, help='2: fused kernel, split bf16 MFMA (default, what the engine runs by default); '
                                                       '1: fused kernel, bit-exact f32 MFMA; 0: one kernel per TF op')
    ap.add_argument('--act', default='selu', help='net_state activation (experiments; the BASELINE config is selu)')
    ap.add_argument('--exchange', choices=['auto', 'full', 'halo', 'slice', 'slice1'], default='auto',
                    help='N > 1: all-gather of whole shards, of boundary rows only (gnn_graph_create_halo), or the feature-sliced '
                         'all-to-all (gnn_loop_set_slice_exchange: every rank aggregates its columns for all nodes).  auto: slice '
                         'from 4 ranks on (2 (P-1)/P^2 instead of (P-1)/P of the state received per iteration), full below.  slice: the '
                         'return all-to-all as ONE grouped call behind the aggregation; slice1: block by block on a second stream beside it '
                         '(gnn_loop_set_slice_exchange(l, 1): verified in loopback groups and over the tests\' stand-in transport, never on '
                         'RCCL with more than one rank - until it has been, the bench keeps to the one-shot form)')
    ap.add_argument('--tile-form', type=int, default=0, help='default path: 1 = one wave per 32-node tile (k_fused), 2 = a wave pair per tile (k_fused_pair), 0 = the library\'s choice')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the untimed configs[1] / configs[4] figures under config.other_configs')
    ap.add_argument('--cpu-iters', type=int, default=30, help='bodies of the CPU-baseline sample = depth of the full-size parity check (default: the workload\'s max_iter)')
    ap.add_argument('--rank-timeout', type=float, default=900.0, help='N > 1: seconds a rank may spend in the timed / validation region before it gives up (exit 124)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:      # bare multi-GPU invocation: become the launcher (no GPU call in this process)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus {args.gpus}` bare (it starts its own ranks) '
                         f'or under torch.distributed.run --nproc-per-node {args.gpus}')

    from GNN import _engine as engine, GNN_utils as utils
    if os.environ.get('GNN_BENCH_ONE_DEVICE') == '1':         # rehearsal on a one-GPU box (tests/test_gpu_multiprocess.py): every rank on device 0,
        local_rank = 0                                        # which RCCL refuses - needs GNN_RCCL_LIBRARY=<the tests' stand-in transport>
    elif engine.device_count() < max(world, local_rank + 1):
        raise SystemExit(f'bench.py --gpus {world}: needs {world} devices, {engine.device_count()} visible (rank {rank})')
    engine.require_device(local_rank)

    d, nl, al, t = 64, 3, 1, 2
    s = utils.syntheticGraph(args.nodes, args.arcs_per_node, nl, al, t, seed=20261003)
    n, e = s['n_nodes'], s['n_arcs']
    rng = np.random.default_rng(20261003)
    st = make_net(rng, al + 2 * (nl + d), [128, 128, d], args.act)
    ou = make_net(rng, nl + d, [t], 'softmax')
    state0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)

    if args.exchange == 'auto':
        args.exchange = 'slice' if (world >= 4 and d % world == 0) else 'full'
    comm, id_path = None, None
    if world > 1:
        uid, id_path = rendezvous_id(rank, world, engine)
        comm = engine.Comm(uid, rank, world, local_rank)
    rb, nr, indptr, adj_src, adj_w, arc_w, arc_lab = engine.shard_csr(n, rank, world, s['indptr'], s['adj_src'], s['adj_w'],
                                                                        s['arc_w'], s['arc_labels_csr'])
    n_arcs_local = len(adj_src)
    if world > 1 and args.exchange == 'halo':
        h = engine.shard_halo(n, rank, world, s['indptr'], s['adj_src'], s['nodes'])
        graph = engine.Graph.halo(n, rank, world, h['block'], h['send_rows'], indptr, h['adj_src'], adj_w, arc_w, arc_lab, h['nodes'],
                                  np.ones(nr, np.uint8), device=local_rank)
    else:
        graph = engine.Graph(n, indptr, adj_src, adj_w, arc_w, arc_lab, s['nodes'], np.ones(nr, np.uint8), row_begin=rb,
                             device=local_rank)
    mst = engine.Mlp(st['weights'], st['activations'], True, device=local_rank)
    mou = engine.Mlp(ou['weights'], ou['activations'], True, device=local_rank)
    loop = engine.Loop(graph, mst, mou, d, args.max_iter, 0.0, comm)
    impl_used = loop.set_impl(args.impl)
    tile_form = loop.set_tile_form(args.tile_form)
    loop.set_state0(state0[rb:rb + nr])
    if world > 1 and args.exchange in ('slice', 'slice1'):
        graph.set_full_adjacency(n, s['indptr'], s['adj_src'], s['adj_w'])
        loop.set_slice_exchange(True, form='pipelined' if args.exchange == 'slice1' else 'oneshot')

    def barrier(value=0.0):
        engine._check(engine.lib().gnn_device_synchronize(local_rank))
        return comm.allreduce_max(value) if comm else value

    with Watchdog(args.rank_timeout if world > 1 else 3600.0, 'warm-up + timed Loops', rank):
        for _ in range(args.warmup):
            loop.run()
        loop.set_profiling(True)
        barrier()
        t0 = time.perf_counter()
        k_total, iter_ms, gap_ms = 0.0, [], []
        for _ in range(args.steps):
            k_total += loop.run()
            tm = loop.timing()
            iter_ms.append(tm['avg_iter_ms'])
            gap_ms.append(tm['avg_between_bodies_ms'])
        elapsed = barrier(time.perf_counter() - t0)          # device sync, then max over ranks
        loop.set_profiling(False)

    # N > 1: the line must validate itself - nobody stands beside a driver's 8-GPU run.  Every rank runs 3 bodies of the SHARDED job and
    # the same 3 bodies UNSHARDED on its own device (the whole graph fits one GPU) and compares its own rows: after three bodies a row
    # depends on rows of every other rank (the synthetic graph has no locality), so a wrong peer, count or offset in any exchange shows
    # in every rank's rows.  The largest difference over all ranks (allreduce max) and the agreement of k go into the line; the exact
    # path (impl 1) must agree bit for bit, and so does the default path (a node's arithmetic does not depend on its tile).
    sharded_check = None
    if world > 1:
        with Watchdog(args.rank_timeout, 'sharded-vs-unsharded validation', rank):
            whole = engine.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8), device=local_rank)
            per_impl, diff_all, k_bad_all = {}, 0.0, 0.0
            for chk_impl in (1, 2):         # the bit-exact path AND the path that was timed
                lp_s = engine.Loop(graph, mst, mou, d, 3, 0.0, comm)
                lp_s.set_impl(chk_impl)
                lp_s.set_state0(state0[rb:rb + nr])
                if args.exchange in ('slice', 'slice1'):
                    lp_s.set_slice_exchange(True, form='pipelined' if args.exchange == 'slice1' else 'oneshot')
                k_s = lp_s.run()
                st_s, out_s = lp_s.state(), lp_s.output()
                lp_s.close()
                lp_u = engine.Loop(whole, mst, mou, d, 3, 0.0)
                lp_u.set_impl(chk_impl)
                lp_u.set_state0(state0)
                k_u = lp_u.run()
                st_u, out_u = lp_u.state()[rb:rb + nr], lp_u.output()[rb:rb + nr]
                lp_u.close()
                diff = max(float(np.max(np.abs(st_s - st_u))) if nr else 0.0, float(np.max(np.abs(out_s - out_u))) if nr else 0.0)
                k_bad = 0.0 if k_s == k_u else 1.0
                d_i, k_i = comm.allreduce_max(diff), comm.allreduce_max(k_bad)
                per_impl[f'impl_{chk_impl}'] = {'sharded_vs_unsharded_max_abs_diff': d_i, 'k_equal': bool(k_i == 0.0)}
                diff_all, k_bad_all = max(diff_all, d_i), max(k_bad_all, k_i)
            whole.close()
            sharded_check = {'bodies': 3, 'impls': [1, 2], 'sharded_vs_unsharded_max_abs_diff': diff_all, 'k_equal': bool(k_bad_all == 0.0),
                             'ok': bool(diff_all == 0.0 and k_bad_all == 0.0), 'per_impl': per_impl,
                             'what': 'every rank: 3 bodies sharded vs the same 3 bodies unsharded on its own device, own rows of state and output, '
                                     'on the bit-exact path and on the default path (a node\'s arithmetic does not depend on its tile or rank: both '
                                     'must agree bit for bit); max over ranks'}
            del st_s, out_s, st_u, out_u

    # the bit-exact fused path (impl 1) on the same inputs, one untimed + one timed Loop: reported beside the headline, and
    # the two final states are compared (the default path must stay within fp32 rounding noise of the exact one)
    exact = None
    if world == 1 and impl_used == 2:
        final_default = loop.state()
        loop.set_impl(1)
        loop.run()
        barrier()
        t2 = time.perf_counter()
        k_x = loop.run()
        barrier()
        dt = time.perf_counter() - t2
        final_exact = loop.state()
        exact = {'updates_per_s': n * k_x / dt, 'ms_per_step': 1e3 * dt,
                 'max_abs_state_difference_to_default_path': float(np.max(np.abs(final_exact - final_default))),
                 'max_abs_state': float(np.max(np.abs(final_exact))),
                 'note': 'random-init weights give an expansive state map: after 30 bodies the exact f32 path itself is about as '
                         'far from float64; one body (below) shows the arithmetic difference proper'}
        loop.set_impl(2)
        one = engine.Loop(graph, mst, mou, d, 1, 0.0)
        one.set_state0(state0)
        one.run()
        s2 = one.state()
        one.set_impl(1)
        one.run()
        exact['max_abs_state_difference_after_one_body'] = float(np.max(np.abs(one.state() - s2)))
        one.close()
        del final_exact, final_default, s2

    # the reference builds the label aggregates (GNN.py:259, :263) in every Loop(); the engine keeps them between Loops until
    # the labels change.  Cold figure: the same Loop with the kept aggregates dropped before each call.
    cold_ms = None
    if world == 1:
        barrier()
        t3 = time.perf_counter()
        for _ in range(2):
            loop.drop_cached_aggregates()
            loop.run()
        barrier()
        cold_ms = 1e3 * (time.perf_counter() - t3) / 2

    # boundary-inclusive rate (never `value`): host state0 in, host state + output back, one Loop (DESIGN.md "Measurement")
    t1 = time.perf_counter()
    loop.set_state0(state0[rb:rb + nr])
    k_e2e = loop.run()
    loop.state(), loop.output()
    e2e_s = time.perf_counter() - t1

    exit_code = 0
    if rank == 0:
        updates = n * k_total
        kernel_ms = float(np.mean(iter_ms))
        alg_bytes = algorithmic_bytes_per_iteration(nr, n_arcs_local, d, nl, al)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel come from a separate rocprofv3 --pmc run of this same command
        # (tools/profile.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), committed under profiles/
        traffic, traffic_src = None, None
        pf = profile_figures() if (world == 1 and impl_used == 2 and args.nodes == 1_000_000) else None
        if pf:
            traffic = pf.get('hbm_bytes_per_launch')
            traffic_src = pf['source'] + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; FETCH_SIZE doubled as the guide prescribes for gfx950)'
        line = {
            'metric': 'node-state-updates/sec (nodes x iters / s), 1M-node synthetic graph',
            'value': updates / elapsed, 'unit': 'node-state-updates/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'synthetic randomGraph-recipe graph, N={n} nodes, E={e} arcs, state_dim=64, '
                                   f'net_state 135->128->128->64 selu+BN, net_output 67->2 softmax+BN, '
                                   f'max_iter={args.max_iter}, threshold=0 (all iterations run); the loop-invariant label aggregates '
                                   f'(GNN.py:259, :263; 0.2 ms) are kept between Loops until the labels change, see cold_aggregates_ms_per_step',
                       'iterations_per_step': k_total / args.steps,
                       'parallelism': (f'node-range shards x{world}, ' + ('RCCL all-to-all of column slices (feature-sliced aggregation' + (', return all-to-all block by block beside the aggregation)' if args.exchange == 'slice1' else ')') if args.exchange in ('slice', 'slice1') else 'RCCL all-gather of ' + ('boundary' if args.exchange == 'halo' else 'owned') + ' state rows') + ' per iteration') if world > 1 else 'single GPU',
                       'impl': {2: 'fused gather+MLP kernel, dense layers on the bf16 MFMA with fp32 operands cut into 3 exact bf16 '
                                   'pieces (6 piece products, fp32 accumulate; error per product <= 3*2^-24)',
                                1: 'fused gather+MLP kernel, dense layers on the f32 MFMA (bit-identical to the oracle)',
                                0: 'one kernel per TF op (unfused)'}[impl_used],
                       'tile_form': {1: 'one wave per 32-node tile (k_fused)', 2: 'wave pair per 32-node tile (k_fused_pair)'}.get(tile_form, 'n/a') if impl_used == 2 else 'n/a',
                       'exact_f32_mfma_path': exact,
                       'cold_aggregates_ms_per_step': cold_ms,
                       'pcie_inclusive_updates_per_s': nr * k_e2e / e2e_s},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'gnn_fused_iteration' if impl_used else 'spmm + dense x3 + check (sum of the per-iteration kernels)',
                         'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_ms': kernel_ms},
        }
        if pf:      # counters of the same kernel from the committed PMC passes (north_star: HBM GB/s and MFMA-busy against gfx950 peak)
            # They are a COPY of profiles/<round>_pmc.json (separate --pmc passes of this command), not measured in this run: they only
            # stand when the profiled build is the one running, i.e. when its launch time matches the live one within 5 %.
            line['roofline']['counters_from_committed_profile'] = True
            prof_ms = pf.get('profiled_avg_launch_ms')
            stale = prof_ms is None or kernel_ms <= 0 or abs(prof_ms - kernel_ms) > 0.05 * kernel_ms
            line['roofline']['profiled_avg_launch_ms'] = prof_ms
            if stale:
                line['roofline']['traffic'] = None
                line['roofline']['counters_dropped'] = (f'profiled launch {prof_ms} ms vs live {kernel_ms:.4f} ms differ by more than 5 %: the committed '
                                                        f'counters describe another build; re-run tools/profile.sh')
            else:
                line['roofline'].update({k: pf[k] for k in ('mfma_busy_pct', 'valu_busy_pct', 'lds_bank_conflict_share', 'valu_insts_per_tile',
                                                            'effective_clock_ghz') if k in pf})
        if world > 1:      # what an iteration of the sharded job is made of, and the self-check of the exchange
            recv = {'full': (world - 1) * (n // world) * d * 4, 'slice': 2 * (world - 1) * (n // world) * (d // world) * 4 if d % world == 0 else None,
                    'slice1': 2 * (world - 1) * (n // world) * (d // world) * 4 if d % world == 0 else None, 'halo': None}[args.exchange]
            line['multi_gpu'] = {'exchange': args.exchange, 'kernel_ms_per_iteration': kernel_ms, 'exchange_ms_per_iteration': float(np.mean(gap_ms)),
                                 'bytes_received_per_rank_per_iteration': recv,
                                 'what': 'rank 0, HIP events on the loop stream: kernel = the fused iteration kernel of the owned rows; exchange = everything '
                                         'between the end of one body\'s kernel and the start of the next (all-gather of rows + flags, or the sliced '
                                         'layout\'s pack / all-to-all / aggregation / all-to-all / unpack)',
                                 'sharded_check': sharded_check}
        if world == 1 and not args.no_other_configs and args.nodes == 1_000_000:
            try:             # untimed extras: whatever happens there must not cost the headline line
                line['config']['other_configs'] = other_configs(engine, s, local_rank)
            except Exception as ex:      # noqa: BLE001
                line['config']['other_configs'] = {'error': repr(ex)}
        parity_ok = None
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(s, st, ou, state0, d, args.cpu_iters, engine, local_rank, full=(graph, mst, mou))
            parity_ok = line['cpu_baseline'].pop('parity_ok', None)
            line['default_path_values'] = line['cpu_baseline'].pop('default_path_values', None)      # what parity_ok says about the TIMED path's values
        if world > 1:
            parity_ok = sharded_check['ok']
        line['parity_ok'] = parity_ok       # null: not checked in this invocation (--no-cpu-baseline)
        print(json.dumps(line), flush=True)
        if parity_ok is False:
            print('bench.py: PARITY FAILED - ' + ('the sharded run differs from the unsharded one' if world > 1 else
                  'the GPU result differs from the oracle at full size (cpu_baseline.gpu_vs_oracle_full_size; default path: default_path_values)'), file=sys.stderr, flush=True)
            exit_code = 3
    if comm:
        barrier()
        if rank == 0 and id_path and os.path.exists(id_path):
            os.remove(id_path)
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == '__main__':
    main()
