"""One rank of tests/test_gpu_multiprocess.py: the engine's multi-process path (gnn_comm_create, the RCCL call sites of loop_exchange /
slice_alltoall / slice_step_aggregate, the sharded readout) with several PROCESSES on one GPU, over the stand-in transport of
tests/mock_rccl (GNN_RCCL_LIBRARY).  Runs every exchange layout on the same small graph and writes k, the rank's state and output rows
per layout to <out>/rank<r>.npz; the test compares them with the C oracle."""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)


def main():
    rank, world, out_dir = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), sys.argv[1]
    import bench
    from GNN import _engine as e
    import test_gpu_sharded as S
    n, d = int(os.environ.get('MP_NODES', 4099)), 8
    g, st, ou, s0 = S._case(4242, n, d, hidden=(16,))
    indptr, adj_src, adj_w, arc_w, arc_lab = S._csr_parts(g)
    mask = np.logical_and(g['set_mask'], g['output_mask'])
    res = {}
    for li, layout in enumerate(('full', 'halo', 'slice1', 'slice2')):
        if layout.startswith('slice') and d % world: continue
        os.environ['GNN_BENCH_RDV'] = os.path.join(out_dir, f'id_{layout}')
        uid, _ = bench.rendezvous_id(rank, world, e)
        comm = e.Comm(uid, rank, world, 0)                       # every rank on device 0
        mst, mou = e.Mlp(st['weights'], st['activations'], st['batch_normalization']), e.Mlp(ou['weights'], ou['activations'], ou['batch_normalization'])
        rb, nr, ip, src, w, aw, al_ = e.shard_csr(n, rank, world, indptr, adj_src, adj_w, arc_w, arc_lab)
        if layout == 'halo':
            h = e.shard_halo(n, rank, world, indptr, adj_src, g['nodes'])
            gr = e.Graph.halo(n, rank, world, h['block'], h['send_rows'], ip, h['adj_src'], w, aw, al_, h['nodes'], mask[rb:rb + nr])
        else:
            gr = e.Graph(n, ip, src, w, aw, al_, g['nodes'], mask[rb:rb + nr], row_begin=rb)
        for impl in (1, 2):
            lp = e.Loop(gr, mst, mou, d, 30, 0.01, comm)
            lp.set_impl(impl)
            lp.set_state0(s0[rb:rb + nr])
            if layout.startswith('slice'):
                gr.set_full_adjacency(n, indptr, adj_src, adj_w)
                lp.set_slice_exchange(True, form='pipelined' if layout[-1] == '1' else 'oneshot')
            k = lp.run()
            k2 = lp.run()                                         # a second Loop on the same communicator
            assert k2 == k
            res[f'{layout}_{impl}_k'] = np.float64(k)
            res[f'{layout}_{impl}_state'] = lp.state()
            res[f'{layout}_{impl}_out'] = lp.output()
            res[f'{layout}_{impl}_max'] = np.float64(comm.allreduce_max(float(rank + 1)))
            lp.close()
        comm.close()
    # ---- the graph readout on shards (GNN.py:331-332; gnn_loop_readout on an RCCL communicator: per-rank partial [G, T], all-gathered, added in
    # rank order) - all-true masks, graphs that straddle shard boundaries among them
    gr_, stg, oug, s0g = S._case(911, 960, 8)
    gr_['set_mask'] = np.ones(960, bool); gr_['output_mask'] = np.ones(960, bool)
    rngg = np.random.default_rng(12)
    bounds = np.sort(rngg.choice(np.arange(1, 960), 11, replace=False))
    sizes = np.diff(np.concatenate([[0], bounds, [960]]))
    ng_indptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ng_node, ng_w = np.arange(960, dtype=np.int32), np.repeat(1.0 / sizes, sizes).astype(np.float32)
    ipg, srcg, wg, awg, alg = S._csr_parts(gr_)
    os.environ['GNN_BENCH_RDV'] = os.path.join(out_dir, 'id_readout')
    uid, _ = bench.rendezvous_id(rank, world, e)
    comm = e.Comm(uid, rank, world, 0)
    rb, nr, ip, src, w, aw, al_ = e.shard_csr(960, rank, world, ipg, srcg, wg, awg, alg)
    grg = e.Graph(960, ip, src, w, aw, al_, gr_['nodes'], np.ones(nr, np.uint8), row_begin=rb)
    lpg = e.Loop(grg, e.Mlp(stg['weights'], stg['activations'], True), e.Mlp(oug['weights'], oug['activations'], True), 8, 20, 0.01, comm)
    lpg.set_impl(1); lpg.set_state0(s0g[rb:rb + nr])
    res['readout_k'] = np.float64(lpg.run())
    res['readout'] = lpg.readout(ng_indptr, ng_node, ng_w)
    lpg.close(); comm.close()

    # ---- LGNN.update_graph across the rank processes (gnn_graph_update_labels on an RCCL communicator: every rank relabels its rows, the
    # label rows are all-gathered - whole shards, or boundary blocks on halo shards), then layer 1 on the relabelled graph
    from util import make_mlp as _mk
    rng1 = np.random.default_rng(31)
    nl1 = g['nodes'].shape[1] + d + 2
    st1 = _mk(rng1, 1 + 2 * (d + nl1), [16, d], 'selu', gain=0.6, bn_random=True)
    ou1 = _mk(rng1, d + nl1, [2], 'softmax', bn_random=True)
    s1 = (0.1 * rng1.standard_normal((n, d))).astype(np.float32)
    for layout in ('full', 'halo'):
        os.environ['GNN_BENCH_RDV'] = os.path.join(out_dir, f'id_lgnn_{layout}')
        uid, _ = bench.rendezvous_id(rank, world, e)
        comm = e.Comm(uid, rank, world, 0)
        m0s, m0o = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
        m1s, m1o = e.Mlp(st1['weights'], st1['activations'], True), e.Mlp(ou1['weights'], ou1['activations'], True)
        rb, nr, ip, src, w, aw, al_ = e.shard_csr(n, rank, world, indptr, adj_src, adj_w, arc_w, arc_lab)
        if layout == 'halo':
            h = e.shard_halo(n, rank, world, indptr, adj_src, g['nodes'])
            base = e.Graph.halo(n, rank, world, h['block'], h['send_rows'], ip, h['adj_src'], w, aw, al_, h['nodes'], mask[rb:rb + nr])
        else:
            base = e.Graph(n, ip, src, w, aw, al_, g['nodes'], mask[rb:rb + nr], row_begin=rb)
        l0 = e.Loop(base, m0s, m0o, d, 20, 0.01, comm)
        l0.set_impl(1); l0.set_state0(s0[rb:rb + nr])
        k0 = l0.run()
        derived = base.derive(d + 2)
        derived.update_labels(base, l0, True, True)
        l1 = e.Loop(derived, m1s, m1o, d, 20, 0.01, comm)
        l1.set_impl(1)
        l1.set_state0(s1[rb:rb + nr])
        k1 = l1.run()
        res[f'lgnn_{layout}_k0'], res[f'lgnn_{layout}_k1'] = np.float64(k0), np.float64(k1)
        res[f'lgnn_{layout}_labels'] = derived.nodes()[:nr] if layout == 'halo' else derived.nodes()[rb:rb + nr]
        res[f'lgnn_{layout}_state'] = l1.state()
        res[f'lgnn_{layout}_out'] = l1.output()
        l1.close(); l0.close(); comm.close()

    # ---- training-mode forward on shards (gnn_loop_train_forward with world > 1): state rows all-gathered after every body, the
    # BatchNormalization statistics and the iteration gates those of all ranks.  Two cases: few rows (per-op kernels) and, MP_TRAIN_WIDE,
    # the wide-layer path.
    from util import make_mlp
    for tag, (nt, dt, hidden, thr_t) in (('train', (1531, 8, (16,), 0.02)), ('trainw', (12000, 64, (128, 128), 0.0))):
        gt, stt, out_, s0t = S._case(777 + nt, nt, dt, hidden=hidden)
        rngt = np.random.default_rng(nt)
        stt = make_mlp(rngt, stt['weights'][0].shape[0], list(hidden) + [dt], 'selu' if tag == 'train' else 'tanh', gain=0.6, bn_random=True)      # (wide case: a smooth activation - SELU's kink makes single gradient entries jump)
        out_ = make_mlp(rngt, out_['weights'][0].shape[0], [2], 'softmax', bn_random=True)
        stt['dropout'], out_['dropout'] = {}, {}
        ipt, srct, wt, awt, alt = S._csr_parts(gt)
        maskt = np.logical_and(gt['set_mask'], gt['output_mask'])
        os.environ['GNN_BENCH_RDV'] = os.path.join(out_dir, f'id_{tag}')
        uid, _ = bench.rendezvous_id(rank, world, e)
        comm = e.Comm(uid, rank, world, 0)
        mst, mou = e.Mlp(stt['weights'], stt['activations'], True), e.Mlp(out_['weights'], out_['activations'], True)
        rb, nr, ip, src, w, aw, al_ = e.shard_csr(nt, rank, world, ipt, srct, wt, awt, alt)
        gr = e.Graph(nt, ip, src, w, aw, al_, gt['nodes'], maskt[rb:rb + nr], row_begin=rb)
        lp = e.Loop(gr, mst, mou, dt, 4, thr_t, comm)
        lp.set_state0(s0t[rb:rb + nr])
        # the arcs that LEAVE the owned rows (by-source CSR of the whole graph, rows rb .. rb + nr; destinations stay global ids)
        from test_gpu_train import _by_source_csr
        sip, sdst, sw = _by_source_csr(gt, nt)
        own = (sip[rb:rb + nr + 1] - sip[rb]).astype(np.int32), sdst[sip[rb]:sip[rb + nr]], sw[sip[rb]:sip[rb + nr]]
        k, outn = lp.train_forward(mst, mou, own, bn_state=np.concatenate(stt['weights'][-4:-2]), bn_output=np.concatenate(out_['weights'][-4:-2]))
        res[f'{tag}_k'] = np.float64(k)
        res[f'{tag}_state'] = lp.state()
        res[f'{tag}_out'] = outn
        # loss over the masked rows of ALL ranks: targets / weights of the whole graph drawn alike on every rank, this rank's rows of them
        m_all = int(maskt.sum())
        rngl = np.random.default_rng(99)
        targets = np.eye(2)[rngl.integers(0, 2, m_all)].astype(np.float32)
        weights = (rngl.uniform(0.5, 1.5, m_all) / m_all).astype(np.float32)
        m0 = int(maskt[:rb].sum())
        loss, d_out = e.loss_grad(0, targets[m0:m0 + outn.shape[0]], outn, weights[m0:m0 + outn.shape[0]])
        back = lp.train_backward(d_out)
        res[f'{tag}_loss'] = np.float64(loss)
        for i, garr in enumerate(back['grads_state']): res[f'{tag}_gs{i}'] = garr
        for i, garr in enumerate(back['grads_output']): res[f'{tag}_go{i}'] = garr
        lp.close(); comm.close()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **res)
    print(f'MP_WORKER_OK rank={rank}')


if __name__ == '__main__':
    main()
