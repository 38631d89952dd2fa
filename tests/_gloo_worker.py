"""Worker of tests/test_sharded_gloo.py (one process per rank, gloo on CPU).

Runs the node-range sharded loop with the SAME plan and exchange protocol as the engine's multi-GPU path
(gnn_shard_range / _engine.shard_csr; per iteration: owned rows computed locally, all-gather of the owned state rows padded
to the shard size + all-gather of the per-rank convergence flag, every rank ORs the flags) but with the oracle's row-wise
arithmetic in place of the HIP kernels, and compares with the single-process oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

from GNN import _engine, GNN_utils as utils      # noqa: E402  (loads libgnn_hip.so; no GPU call is made)
from oracle import c_oracle as corc              # noqa: E402
from oracle import gnn_oracle as orc             # noqa: E402
from util import make_mlp                        # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    n, d, nl, al, max_it, thr = 1000, 8, 3, 1, 25, 0.01
    s = utils.syntheticGraph(n, 8.0, nl, al, 2, seed=77)
    rng = np.random.default_rng(5)
    st = make_mlp(rng, al + 2 * (nl + d), [16, d], 'selu', gain=0.6)
    ou = make_mlp(rng, nl + d, [2], 'softmax')
    state0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)

    rb, nr, indptr, adj_src, adj_w, arc_w, arc_lab = _engine.shard_csr(n, rank, world, s['indptr'], s['adj_src'], s['adj_w'],
                                                                        s['arc_w'], s['arc_labels_csr'])
    shard = ((n + world - 1) // world + 31) // 32 * 32
    assert rb == min(n, shard * rank) and nr == min(n, shard * (rank + 1)) - rb
    ranges = [_engine.shard_range(n, r, world) for r in range(world)]
    assert sum(c for _, c in ranges) == n and all(ranges[i][0] + ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))

    local = (indptr, adj_src, adj_w)
    agg_nodes = corc.spmm(local, s['nodes'])                                         # GNN.py:263 on the owned rows
    agg_arcs = corc.spmm((indptr, np.arange(len(arc_w), dtype=np.int32), arc_w), arc_lab)   # GNN.py:259

    def any_flag(flag):
        out = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(out, torch.tensor([int(flag)], dtype=torch.int32))
        return any(int(t.item()) for t in out)

    # padded replica -> global rows: shard p holds rows [p*shard, p*shard + count_p)
    def to_global(full_padded):
        return np.concatenate([full_padded[p * shard:p * shard + ranges[p][1]] for p in range(world)])

    def gather_state(own):
        buf = np.zeros((shard, own.shape[1]), np.float32)
        buf[:nr] = own
        out = [torch.zeros(shard, own.shape[1]) for _ in range(world)]
        dist.all_gather(out, torch.from_numpy(buf))
        return to_global(np.concatenate([t.numpy() for t in out]))

    state = gather_state(state0[rb:rb + nr])
    assert np.array_equal(state, state0)
    go = any_flag(orc.not_converged(state[rb:rb + nr], np.ones((nr, d), np.float32), thr).any())
    k = 0
    while go and k < max_it:
        own = state[rb:rb + nr]
        inp = np.concatenate([own, s['nodes'][rb:rb + nr], corc.spmm(local, state), agg_nodes, agg_arcs], axis=1)
        new = corc.mlp_forward(inp, st['weights'], st['activations'], True)
        flag = orc.not_converged(new, own, thr).any()
        state = gather_state(new)
        go = any_flag(flag)
        k += 1

    # ---- the feature-sliced protocol (gnn_loop_set_slice_exchange) with the same arithmetic: rank q aggregates columns
    # [q d/P, (q+1) d/P) of ALL nodes over the whole graph's adjacency; two all-to-all steps per iteration; no rank ever holds
    # another rank's state rows.  Must give the same bits (the fmaf chain of an aggregated element is the same).
    k_sl, state_sl = None, None
    if d % world == 0:
        cs = d // world
        full = (s['indptr'], s['adj_src'], s['adj_w'])

        def alltoall(blocks):                                   # blocks[q]: [shard, cs] for rank q -> list of what every rank sent me
            # gloo has no all_to_all: every rank all-gathers its P blocks and keeps the ones addressed to it
            mine = []
            gathered = [torch.zeros(world, shard, cs) for _ in range(world)]
            dist.all_gather(gathered, torch.from_numpy(np.stack(blocks).astype(np.float32)))
            for p_ in range(world):
                mine.append(gathered[p_][rank].numpy())
            return mine

        def pad(rows):                                          # [<= shard, w] -> [shard, w]
            buf = np.zeros((shard, rows.shape[1]), np.float32)
            buf[:rows.shape[0]] = rows
            return buf

        def return_pipelined(slice_all):
            # the schedule of slice_step_aggregate (csrc/gnn_engine.hip, gnn_loop_set_slice_exchange(l, 1)): the slice is aggregated in one
            # row block per destination rank in the order rank + 1, ..., rank; at step t rank r sends block (r + 1 + t) % P to that rank
            # and receives its own rows' block from rank (r - 1 - t) % P - a permutation per step; the last step is the rank's own block
            back = [None] * world
            for t in range(world):
                q, frm = (rank + 1 + t) % world, (rank - 1 - t) % world
                r0, cnt = ranges[q]
                rows_ip = full[0][r0:r0 + cnt + 1]
                block = np.zeros((cnt, cs), np.float32)
                if cnt:      # rows r0 .. r0 + cnt of the whole graph's CSR (absolute arc offsets, like indptr + r0 on the device)
                    sub = (rows_ip - rows_ip[0], full[1][rows_ip[0]:rows_ip[-1]], full[2][rows_ip[0]:rows_ip[-1]])
                    block = corc.spmm(sub, np.ascontiguousarray(slice_all))
                if q == rank:
                    assert frm == rank
                    back[rank] = pad(block)
                else:
                    recv = torch.zeros(shard, cs)
                    reqs = [dist.isend(torch.from_numpy(pad(block)), q), dist.irecv(recv, frm)]
                    for r_ in reqs: r_.wait()
                    back[frm] = recv.numpy()
            return back

        own = state0[rb:rb + nr].copy()
        go = any_flag(orc.not_converged(own, np.ones((nr, d), np.float32), thr).any())
        k_sl = 0
        while go and k_sl < max_it:
            sent = alltoall([pad(own[:, q * cs:(q + 1) * cs]) for q in range(world)])       # my column slice of every rank's rows
            slice_all = to_global(np.concatenate(sent))                                     # [n, cs]
            agg_slice = corc.spmm(full, np.ascontiguousarray(slice_all))                     # [n, cs]: all nodes, my columns
            back = alltoall([pad(agg_slice[ranges[q][0]:ranges[q][0] + ranges[q][1]]) for q in range(world)])       # rank q's rows, my columns
            back_p = return_pipelined(slice_all)                                             # the same through the block-by-block schedule
            assert all(np.array_equal(x, y) for x, y in zip(back, back_p)), 'pipelined return all-to-all differs from the grouped one'
            agg_own = np.concatenate([b[:nr] for b in back], axis=1)                         # [nr, d]: my rows, all columns
            inp = np.concatenate([own, s['nodes'][rb:rb + nr], agg_own, agg_nodes, agg_arcs], axis=1)
            new = corc.mlp_forward(inp, st['weights'], st['activations'], True)
            go = any_flag(orc.not_converged(new, own, thr).any())
            own = new
            k_sl += 1
        state_sl = gather_state(own)

    # single-process oracle on the whole graph
    arcs = np.concatenate([np.stack([s['src'], s['dst']], 1).astype(np.float32), s['arc_labels']], axis=1)
    g = dict(nodes=s['nodes'], arcs=arcs, set_mask=np.ones(n, bool), output_mask=np.ones(n, bool),
             adjT=(s['indptr'], s['adj_src'], s['adj_w']), arcT=(s['indptr'], s['arc_perm'], s['arc_w']))
    kc, sc, _ = corc.loop_node(g, st, ou, d, max_it, thr, state0)
    ks = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(ks, torch.tensor([k], dtype=torch.int32))
    assert len({int(t.item()) for t in ks}) == 1, 'ranks disagree on the iteration count'
    assert k == kc and 1 < k < max_it, (k, kc)
    assert np.array_equal(state, sc), 'sharded state differs from the single-process oracle'
    if state_sl is not None:
        assert k_sl == kc and np.array_equal(state_sl, sc), 'feature-sliced protocol differs from the single-process oracle'
    dist.barrier()
    if rank == 0:
        print(f'SHARDED_OK world={world} k={k} sliced={state_sl is not None}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
