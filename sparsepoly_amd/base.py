"""Estimator base classes -- mirror of reference ``sparsepoly/base.py``."""
from abc import ABCMeta

import numpy as np
from sklearn.base import BaseEstimator, ClassifierMixin, RegressorMixin
from sklearn.preprocessing import LabelBinarizer
from sklearn.utils.multiclass import type_of_target
from sklearn.utils.validation import check_X_y

from .loss import CLASSIFICATION_LOSSES, REGRESSION_LOSSES


class BaseSparsePoly(BaseEstimator, metaclass=ABCMeta):
    def _get_loss(self, loss):
        """base.py:18-25"""
        if loss not in self._LOSSES:
            losses_str = '", "'.join(self._LOSSES)
            raise ValueError(
                f"Loss function {loss} not supported. The available options are:"
                f' "{losses_str}".'
            )
        return self._LOSSES[loss]

    def _get_regularizer(self, regularizer):
        """base.py:27-34"""
        if regularizer not in self._REGULARIZERS:
            regularizers_str = '", "'.join(self._REGULARIZERS)
            raise ValueError(
                f"Regularizer {regularizer} not supported. The available options are:"
                f' "{regularizers_str}".'
            )
        return self._REGULARIZERS[regularizer]()


class SparsePolyRegressorMixin(RegressorMixin):
    _LOSSES = REGRESSION_LOSSES

    def _check_X_y(self, X, y):
        """base.py:40-50"""
        X, y = check_X_y(X, y, accept_sparse=True, multi_output=False, dtype=np.double,
                         y_numeric=True)
        y = y.astype(np.double).ravel()
        return X, y

    def predict(self, X):
        """base.py:52-65"""
        return self._predict(X)


class SparsePolyClassifierMixin(ClassifierMixin):
    _LOSSES = CLASSIFICATION_LOSSES

    def decision_function(self, X):
        """base.py:70-84"""
        return self._predict(X)

    def predict(self, X):
        """base.py:86-100"""
        y_pred = self.decision_function(X) > 0
        return self.label_binarizer_.inverse_transform(y_pred)

    def predict_proba(self, X):
        """base.py:102-124"""
        if self.loss == "logistic":
            return 1 / (1 + np.exp(-self.decision_function(X)))
        else:
            raise ValueError(
                "Probability estimates only available for "
                "loss='logistic'. You may use probability "
                "calibration methods from scikit-learn instead."
            )

    def _check_X_y(self, X, y):
        """base.py:126-142"""
        is_2d = hasattr(y, "shape") and len(y.shape) > 1 and y.shape[1] >= 2
        if is_2d or type_of_target(y) != "binary":
            raise TypeError(
                "Only binary targets supported. For training "
                "multiclass or multilabel models, you may use the "
                "OneVsRest or OneVsAll metaestimators in "
                "scikit-learn."
            )
        X, Y = check_X_y(X, y, dtype=np.double, accept_sparse=True, multi_output=False)
        self.label_binarizer_ = LabelBinarizer(pos_label=1, neg_label=-1)
        y = self.label_binarizer_.fit_transform(Y).ravel().astype(np.double)
        return X, y
