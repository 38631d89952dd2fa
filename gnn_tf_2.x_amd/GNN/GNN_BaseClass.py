"""Common base of GNN / LGNN with the reference's public surface (``GNN/GNN_BaseClass.py``): ``evaluate``, ``test``,
``train``, ``LKO``, ``checktype``, ``get_filtered_tensor`` and the ``history`` helpers.

Everything here is host-side orchestration around ``Loop`` (which runs on the MI355X).  Differences from the reference,
all deliberate:
  * the constructor does NOT delete ``path_writer`` (reference GNN_BaseClass.py:58 rmtree's it); call ``clear_writer()``;
  * ``train`` computes its gradients on the device (``gnn_loop_train_step`` / ``gnn_loop_train_forward`` + ``_backward``:
    training-mode forward, loss, back-propagation through the unrolled loop) and applies the optimizer on the host;
    TensorBoard summaries are not written.  Nothing falls back to a CPU.
"""
from __future__ import annotations

import os
import shutil
from abc import ABC, abstractmethod
from typing import Optional, Union

import numpy as np

from GNN.graph_class import GraphObject, GraphTensor


class BaseClass(ABC):
    def __init__(self, optimizer, loss_function, loss_arguments: Optional[dict], addressed_problem: str,
                 extra_metrics: Optional[dict] = None, extra_metrics_arguments: Optional[dict[str, dict]] = None,
                 path_writer: str = 'writer/', namespace='GNN') -> None:
        if addressed_problem not in ['c', 'r']: raise ValueError('param <addressed_problem> not in [\'c\',\'r\']')
        if not isinstance(extra_metrics, (dict, type(None))): raise TypeError('type of param <extra_metrics> must be None or dict')
        self.loss_function = loss_function
        self.loss_args = dict() if loss_arguments is None else loss_arguments
        self.optimizer = optimizer
        self.addressed_problem = addressed_problem
        self.extra_metrics = dict() if extra_metrics is None else extra_metrics
        self.mt_args = dict() if extra_metrics_arguments is None else extra_metrics_arguments
        if path_writer[-1] != '/': path_writer += '/'
        self.path_writer = path_writer
        self.namespace = namespace if isinstance(namespace, list) else [namespace]
        self.history = dict()

    def clear_writer(self) -> None:
        """Explicit form of what the reference constructor does implicitly (GNN_BaseClass.py:58)."""
        if os.path.exists(self.path_writer): shutil.rmtree(self.path_writer)

    # ---- to be provided by GNN / LGNN ----------------------------------------------------------------------------------
    @abstractmethod
    def copy(self, *, path_writer: str = '', namespace: str = '', copy_weights: bool = True): ...

    @abstractmethod
    def get_weights(self): ...

    @abstractmethod
    def set_weights(self, weights_state, weights_output) -> None: ...

    @abstractmethod
    def Loop(self, g, *, training: bool = False): ...

    @abstractmethod
    def __call__(self, g): ...

    @abstractmethod
    def evaluate_single_graph(self, g, training: bool) -> tuple: ...

    # ---- history ---------------------------------------------------------------------------------------------------------
    def printHistory(self) -> None:
        from pandas import DataFrame
        print('\n', DataFrame(self.history), end='\n\n')

    def saveHistory_csv(self, path) -> None:
        from pandas import DataFrame
        if path[-3:] != '.csv': path += '.csv'
        DataFrame(self.history).to_csv(path, index=False)

    def saveHistory_txt(self, path) -> None:
        from pandas import DataFrame
        if path[-3:] != '.txt': path += '.txt'
        with open(path, 'w') as txt:
            txt.write(DataFrame(self.history).to_string(index=False))

    # ---- evaluation (reference GNN_BaseClass.py:165-189, 338-359) -----------------------------------------------------
    def evaluate(self, g) -> tuple:
        """metrics dict (+ 'It', 'Loss'), y_true, y_pred, targets, y_score over one graph or a list of graphs."""
        graphs = self.checktype(g)
        marked = self._prerun(graphs) if hasattr(self, '_prerun') else []       # (GNN node- / graph-based: the graphs' Loops in one call, small ones side by side)
        try:
            iters, losses, targets, outs = zip(*[self.evaluate_single_graph(i, training=False) for i in graphs])
        finally:
            for lp in marked: lp._fresh = None
        targets = np.concatenate(targets, axis=0)
        y_score = np.concatenate(outs, axis=0)
        classify = self.addressed_problem == 'c'
        y_true = np.argmax(targets, axis=1) if classify else targets
        y_pred = np.argmax(y_score, axis=1) if classify else y_score
        metrics = {k: float(np.mean(f(y_true, y_pred, **self.mt_args.get(k, dict())))) for k, f in self.extra_metrics.items()}
        metrics['It'] = int(np.mean(np.asarray(iters, dtype=np.float32)))     # int(mean) as in the reference (:187)
        metrics['Loss'] = float(np.mean(np.asarray(losses, dtype=np.float32)))
        return metrics, y_true, y_pred, targets, y_score

    def test(self, gTe, *, rocdir: str = '', micro_and_macro: bool = False, prisofsdir: str = '', pos_label=0) -> dict:
        metrics, y_true, y_pred, targets, y_score = self.evaluate(self.checktype(gTe))
        if rocdir or prisofsdir:
            import GNN.GNN_metrics as mt
            mt.ROC(targets, y_score, rocdir, micro_and_macro, pos_label=pos_label)
        return metrics

    def training_step(self, g: GraphTensor, mean: bool) -> None:
        """Gradients of one batch and the optimizer update (reference GNN_BaseClass.py:231-247); provided by GNN / LGNN."""
        raise NotImplementedError(f'{type(self).__name__}.train(): back-propagation for this model type is not implemented '
                                  f'on the MI355X engine yet')

    def train(self, gTr, epochs: int, gVa=None, update_freq: int = 10, max_fails: int = 10, observed_metric='Loss', policy='min',
              *, mean: bool = True, verbose: int = 3) -> None:
        """Learning procedure with the reference's signature and bookkeeping (GNN_BaseClass.py:192-335): one training_step per
        batch and epoch; every ``update_freq`` epochs the training (and validation) sets are evaluated into ``self.history``;
        early stopping on ``observed_metric`` of gVa after ``max_fails`` evaluations without improvement, restoring the best
        weights.  TensorBoard summaries of the reference are not written."""
        if verbose not in range(4): raise ValueError('param <verbose> not in [0,1,2,3]')
        gTr = self.checktype(gTr)
        gVa = self.checktype(gVa)
        if not self.history:
            keys = ['Epoch'] + [i + j for i in ['It', 'Loss'] + list(self.extra_metrics) for j in ([' Tr', ' Va'] if gVa else [' Tr'])]
            if gVa: keys += ['Fail', f'Best {observed_metric} Va']
            self.history.update({i: list() for i in keys})
        best_key = f'Best {observed_metric} Va'
        if gVa:
            assert policy in ['min', 'max']
            better = np.less if policy == 'min' else np.greater
            best = self.history[best_key][-1] if self.history[best_key] else (float(1e30) if policy == 'min' else float(-1e30))
            fails, best_ws, best_wo = 0, *self.get_weights()
        first = self.history['Epoch'][-1] + 1 if self.history['Epoch'] else 0
        last = first + epochs
        stopped = False
        for e in range(first, last):
            for i, batch in enumerate(gTr):
                self.training_step(batch, mean=mean)
                if verbose > 2: print(f' > Epoch {e:4d}/{last} \t\t> Batch {i + 1:4d}/{len(gTr)}', end='\r')
            if e % update_freq == 0:
                metrics_tr, *_ = self.evaluate(gTr)
                self.history['Epoch'].append(e)
                for key, val in metrics_tr.items(): self.history[f'{key} Tr'].append(val)
                if gVa:
                    metrics_va, *_ = self.evaluate(gVa)
                    value = metrics_va[observed_metric]
                    if better(value, best):
                        best, fails = value, 0
                        best_ws, best_wo = self.get_weights()
                    else:
                        fails += 1
                    self.history[best_key].append(best)
                    self.history['Fail'].append(fails)
                    for key, val in metrics_va.items(): self.history[f'{key} Va'].append(val)
                    if fails >= max_fails:
                        if verbose in [1, 3]: self.printHistory()
                        print('\r Validation Stop')
                        stopped = True
                        break
                if verbose in [1, 3]: self.printHistory()
        if not stopped: print('\r End of Epochs Stop')
        if gVa: self.set_weights(best_ws, best_wo)

    def LKO(self, batches, epochs: int = 500, training_mode=None, update_freq: int = 10, max_fails: int = 10,
            observed_metric: str = 'Loss', policy='min', mean: bool = True, verbose: int = 3) -> dict:
        """Leave-K-out driver (reference GNN_BaseClass.py:362-402); needs train()."""
        metrics = {i: list() for i in list(self.extra_metrics) + ['It', 'Loss']}
        kwargs = {'training_mode': training_mode} if training_mode else {}
        n = len(batches[0])
        for i, (gTr, gTe, gVa) in enumerate(zip(*batches)):
            print(f'\nBATCH K-OUT {i + 1}/{n}')
            temp = self.copy(copy_weights=False, path_writer=f'{self.path_writer}{i}', namespace=f'Batch {i + 1}-{n}')
            temp.train(gTr, epochs, gVa, update_freq, max_fails, observed_metric, policy, mean=mean, verbose=verbose, **kwargs)
            for name, value in temp.test(gTe).items(): metrics[name].append(value)
        return metrics

    # ---- static helpers -------------------------------------------------------------------------------------------------
    @staticmethod
    def get_filtered_tensor(g: GraphTensor, inp):
        """targets / sample_weights of the nodes that are in the set (reference GNN_BaseClass.py:405-410):
        inp is indexed by output_mask==True positions, of which set_mask selects a subset."""
        return np.asarray(inp)[np.asarray(g.set_mask)[np.asarray(g.output_mask)]]

    @staticmethod
    def checktype(elem):
        """None, or a list of GraphTensor (reference GNN_BaseClass.py:413-425)."""
        if elem is None:
            return None
        if isinstance(elem, GraphTensor):
            return [elem]
        if isinstance(elem, GraphObject):
            return [GraphTensor.fromGraphObject(elem)]
        if isinstance(elem, (list, tuple)) and all(isinstance(g, (GraphObject, GraphTensor)) for g in elem):
            return [GraphTensor.fromGraphObject(g) if isinstance(g, GraphObject) else g for g in elem]
        raise TypeError('Error - <gTr> and/or <gVa> are not GraphObject/GraphTensor or LIST/TUPLE of GraphObjects/GraphTensors')
