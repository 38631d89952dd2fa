"""TensorFlow-free counterpart of the reference's starter.py: same module-level configuration names and the same result
names (`graphs, gTr, gVa, gTe, gnn, lgnn`), models running on the MI355X engine.

    cd gnn_tf_2.x_amd && python -i starter.py
    >>> gnn.test(gTe)            # forward Loop / evaluate / test run on the GPU
    >>> lgnn.test(gTe)
    >>> gnn.train(gTr, 20, gVa)  # gradients on the GPU (gnn_loop_train_step), Adam on the host
`lgnn.train` supports training_mode='serial' | 'parallel' | 'residual' (reference LGNN.py:293-344).
"""
from __future__ import annotations

import os
import sys
from typing import Optional, Union

from numpy import random

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from GNN import GNN_metrics as mt, GNN_utils as utils, losses, optimizers
from GNN.GNN import GNNnodeBased, GNNedgeBased, GNNgraphBased
from GNN.LGNN import LGNN
from GNN.MLP import MLP, get_inout_dims
from GNN.graph_class import GraphObject

# ---- script options (same names and defaults as reference starter.py:23-86) --------------------------------------------
use_MUTAG: bool = True
problem_based: str = 'n'
addressed_problem: str = 'c'
graphs_number: int = 100
min_nodes_number: int = 15
max_nodes_number: int = 40
dim_node_label: int = 3
dim_arc_label: int = 1
dim_target: int = 2
density: float = 0.7
aggregation_mode: str = 'average'

perc_Train: float = 0.7
perc_Valid: float = 0.2
batch_size: int = 32
normalize: bool = True
seed: Optional[int] = None
norm_nodes_range: Optional[tuple] = None
norm_arcs_range: Optional[tuple] = None

activations_net_state: str = 'selu'
kernel_init_net_state: str = 'lecun_normal'
bias_init_net_state: str = 'lecun_normal'
dropout_rate_st: float = 0.1
dropout_pos_st: Union[list, int] = 0
hidden_units_net_state: Optional[Union[list, int]] = None

activations_net_output: str = 'softmax'
kernel_init_net_output: str = 'glorot_normal'
bias_init_net_output: str = 'glorot_normal'
dropout_rate_out: float = 0.1
dropout_pos_out: Union[list, int] = 0
hidden_units_net_output: Optional[Union[list, int]] = None

dim_state: int = 0
max_iter: int = 5
state_threshold: float = 0.01

layers: int = 5
get_state: bool = False
get_output: bool = True
path_writer: str = 'writer/'
optimizer = optimizers.Adam(learning_rate=0.001)
lossF = losses.categorical_crossentropy
lossArguments: Optional[dict] = {'from_logits': False}
extra_metrics: Optional[dict] = {i: mt.Metrics[i] for i in ['Acc', 'Bacc', 'Tpr', 'Tnr', 'Fpr', 'Fnr', 'Ck', 'Js', 'Prec', 'Rec', 'Fs']}
metrics_args: Optional[dict] = {i: {'average': 'weighted', 'zero_division': 0} for i in ['Fs', 'Prec', 'Rec', 'Js']}

# ---- dataset -------------------------------------------------------------------------------------------------------------
if use_MUTAG:
    addressed_problem, problem_based = 'c', 'g'
    from load_MUTAG import load as _load_mutag
    graphs = _load_mutag()
else:
    graphs = [utils.randomGraph(nodes_number=int(random.choice(range(min_nodes_number, max_nodes_number))), dim_node_label=dim_node_label,
                                dim_arc_label=dim_arc_label, dim_target=dim_target, density=density, normalize_features=False,
                                aggregation_mode=aggregation_mode, problem_based=problem_based) for _ in range(graphs_number)]

iTr, iTe, iVa = utils.getindices(len(graphs), perc_Train, perc_Valid, seed=seed)
gTr = utils.getbatches([graphs[i] for i in iTr], batch_size=batch_size, problem_based=problem_based, aggregation_mode=aggregation_mode)
gVa = GraphObject.merge([graphs[i] for i in iVa], problem_based=problem_based, aggregation_mode=aggregation_mode)
gTe = GraphObject.merge([graphs[i] for i in iTe], problem_based=problem_based, aggregation_mode=aggregation_mode)
gGen = gTr[0].copy()
if normalize:
    utils.normalize_graphs(gTr, gVa, gTe, based_on='gTr', norm_rangeN=norm_nodes_range, norm_rangeA=norm_arcs_range)

# ---- models ----------------------------------------------------------------------------------------------------------------
def _nets(net_name, hidden, activations, kinit, binit, drate, dpos):
    dims = [get_inout_dims(net_name=net_name, dim_node_label=gGen.DIM_NODE_LABEL, dim_arc_label=gGen.DIM_ARC_LABEL,
                           dim_target=gGen.DIM_TARGET, problem_based=problem_based, dim_state=dim_state, hidden_units=hidden,
                           layer=i, get_state=get_state, get_output=get_output) for i in range(layers)]
    # MLP's default BatchNormalization after the LAST layer applies to net_output too, as in the reference (MLP.py:13, :63): with a
    # 2-class softmax the two normalised columns are exact opposites, categorical_crossentropy then renormalises by a sum of ~0 and
    # clips, and no gradient reaches the weights - the reference's default behaves the same.  GNN_STARTER_OUTPUT_BN=0 turns that
    # layer off for net_output (what tools/run_starter.py does to show the training loop learning).
    bn = net_name != 'output' or os.environ.get('GNN_STARTER_OUTPUT_BN', '1') != '0'
    return [MLP(input_dim=i, layers=j, activations=activations, kernel_initializer=kinit, bias_initializer=binit,
                dropout_rate=drate, dropout_pos=dpos, batch_normalization=bn) for i, j in dims]

nets_St = _nets('state', hidden_units_net_state, activations_net_state, kernel_init_net_state, bias_init_net_state, dropout_rate_st, dropout_pos_st)
nets_Out = _nets('output', hidden_units_net_output, activations_net_output, kernel_init_net_output, bias_init_net_output, dropout_rate_out, dropout_pos_out)

gnntype = {'n': GNNnodeBased, 'a': GNNedgeBased, 'g': GNNgraphBased}[problem_based]
gnns = [gnntype(net_state=st, net_output=out, optimizer=optimizer.__class__(**optimizer.get_config()), loss_function=lossF, loss_arguments=lossArguments,
                state_vect_dim=dim_state, max_iteration=max_iter, threshold=state_threshold, addressed_problem=addressed_problem,
                extra_metrics=extra_metrics, extra_metrics_arguments=metrics_args, path_writer=f'{path_writer}/GNN{idx}')
        for idx, (st, out) in enumerate(zip(nets_St, nets_Out))]
gnn = gnns[0].copy(path_writer=f'{path_writer}GNN_single', copy_weights=True)
lgnn = LGNN(gnns=gnns, get_state=get_state, get_output=get_output, optimizer=optimizer, loss_function=lossF,
            loss_arguments=lossArguments, addressed_problem=addressed_problem, extra_metrics=extra_metrics,
            extra_metrics_arguments=metrics_args, path_writer=f'{path_writer}LGNN', namespace='LGNN')
