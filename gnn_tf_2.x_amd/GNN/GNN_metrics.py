# coding=utf-8
"""Host-side metric table with the reference's keys (GNN/GNN_metrics.py:152).  sklearn wrappers on small arrays; the ROC /
precision-recall plotting of the reference is outside the hot path and not provided."""
import numpy as np
from sklearn import metrics as mt


def TPR(y_true, y_pred): return mt.recall_score(y_true=y_true, y_pred=y_pred)
def TNR(y_true, y_pred): return 2 * mt.balanced_accuracy_score(y_true=y_true, y_pred=y_pred) - TPR(y_true, y_pred)
def FPR(y_true, y_pred): return 1 - TNR(y_true, y_pred)
def FNR(y_true, y_pred): return 1 - TPR(y_true, y_pred)


def accuracy_per_class(y_true, y_pred, class_label: int = None):
    cm = mt.confusion_matrix(y_true=y_true, y_pred=y_pred)
    acc = np.diag(cm) / cm.sum(axis=1)
    return acc if class_label is None else acc[class_label]


def ROC(*_, **__):
    raise NotImplementedError('ROC plotting is not part of the MI355X engine (host-side matplotlib code in the reference)')


PRISOFS = ROC

Metrics = {'Acc': mt.accuracy_score, 'Bacc': mt.balanced_accuracy_score, 'Js': mt.jaccard_score, 'Ck': mt.cohen_kappa_score,
           'Prec': mt.precision_score, 'Rec': mt.recall_score, 'Fs': mt.f1_score, 'Tpr': TPR, 'Tnr': TNR, 'Fpr': FPR,
           'Fnr': FNR, 'Cl0': accuracy_per_class, 'Cl1': accuracy_per_class}
