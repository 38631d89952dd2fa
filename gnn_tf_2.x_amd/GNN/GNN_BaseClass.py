"""Common base of GNN / LGNN with the reference's public surface (``GNN/GNN_BaseClass.py``): ``evaluate``, ``test``,
``train``, ``LKO``, ``checktype``, ``get_filtered_tensor`` and the ``history`` helpers.

Everything here is host-side orchestration around ``Loop`` (which runs on the MI355X).  Differences from the reference,
all deliberate:
  * the constructor does NOT delete ``path_writer`` (reference GNN_BaseClass.py:58 rmtree's it); call ``clear_writer()``;
  * ``train`` needs back-propagation through the unrolled loop, which the device engine does not provide yet: it raises
    ``NotImplementedError`` instead of silently training on a CPU fallback.
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
        iters, losses, targets, outs = zip(*[self.evaluate_single_graph(i, training=False) for i in graphs])
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

    def train(self, gTr, epochs: int, gVa=None, update_freq: int = 10, max_fails: int = 10, observed_metric='Loss', policy='min',
              *, mean: bool = True, verbose: int = 3) -> None:
        """Same signature as reference GNN_BaseClass.py:192-195."""
        if verbose not in range(4): raise ValueError('param <verbose> not in [0,1,2,3]')
        self.checktype(gTr), self.checktype(gVa)
        raise NotImplementedError('train(): back-propagation through the unrolled state loop (reference GNN_BaseClass.py:231-247) '
                                  'is not implemented on the MI355X engine yet; forward Loop/evaluate/test are')

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
