"""Input validation and the regressor / classifier faces of the estimators.

Behavioural contract (reference ``sparsepoly/base.py:17-142``): same class names, same error
types and texts, same target handling (regressors: numeric 1-d targets cast to float64;
classifiers: binary targets only, mapped to -1 / +1 by a ``LabelBinarizer`` kept as
``label_binarizer_``).  The code is organised around two module-level helpers -- a registry
look-up and a target validator -- with the mixins as thin shells.
"""
import numpy as np
from sklearn.base import BaseEstimator, ClassifierMixin, RegressorMixin
from sklearn.preprocessing import LabelBinarizer
from sklearn.utils.multiclass import type_of_target
from sklearn.utils.validation import check_X_y

from .loss import CLASSIFICATION_LOSSES, REGRESSION_LOSSES

_NOT_BINARY = ("Only binary targets supported. For training multiclass or multilabel models, "
               "you may use the OneVsRest or OneVsAll metaestimators in scikit-learn.")
_NO_PROBA = ("Probability estimates only available for loss='logistic'. You may use "
             "probability calibration methods from scikit-learn instead.")


def _registry_lookup(kind, name, table):
    """``table[name]`` or the reference's ValueError listing the valid names
    (``base.py:19-24`` for losses, ``:28-33`` for regularizers)."""
    try:
        return table[name]
    except KeyError:
        options = '", "'.join(table)
        raise ValueError('%s %s not supported. The available options are: "%s".'
                         % (kind, name, options)) from None


def _validated_xy(X, y, binary):
    """``check_X_y`` as the reference calls it (``base.py:41-49,138``); for ``binary`` the
    target must be a 1-d two-class vector (``:130-136``).  Returns X, y and -- for binary
    targets -- the fitted ``LabelBinarizer``."""
    if binary:
        two_d = getattr(y, "ndim", None) is not None and np.ndim(y) > 1 and np.shape(y)[1] >= 2
        if two_d or type_of_target(y) != "binary":
            raise TypeError(_NOT_BINARY)
        X, y = check_X_y(X, y, dtype=np.double, accept_sparse=True, multi_output=False)
        binarizer = LabelBinarizer(pos_label=1, neg_label=-1)
        return X, binarizer.fit_transform(y).ravel().astype(np.double), binarizer
    X, y = check_X_y(X, y, accept_sparse=True, multi_output=False, dtype=np.double,
                     y_numeric=True)
    return X, np.asarray(y, dtype=np.double).ravel(), None


class BaseSparsePoly(BaseEstimator):
    """Registry access shared by every estimator (``_LOSSES`` / ``_REGULARIZERS`` are class
    attributes of the concrete estimators)."""

    def _get_loss(self, loss):
        return _registry_lookup("Loss function", loss, self._LOSSES)

    def _get_regularizer(self, regularizer):
        return _registry_lookup("Regularizer", regularizer, self._REGULARIZERS)()


class SparsePolyRegressorMixin(RegressorMixin):
    _LOSSES = REGRESSION_LOSSES

    def _check_X_y(self, X, y):
        return _validated_xy(X, y, binary=False)[:2]

    def predict(self, X):
        """Predicted targets, shape (n_samples,) (``base.py:52-65``)."""
        return self._predict(X)


class SparsePolyClassifierMixin(ClassifierMixin):
    _LOSSES = CLASSIFICATION_LOSSES

    def _check_X_y(self, X, y):
        X, y, self.label_binarizer_ = _validated_xy(X, y, binary=True)
        return X, y

    def decision_function(self, X):
        """Raw model output; positive means the positive class (``base.py:70-84``)."""
        return self._predict(X)

    def predict(self, X):
        """Class labels as seen in ``fit`` (``base.py:86-100``)."""
        return self.label_binarizer_.inverse_transform(self.decision_function(X) > 0)

    def predict_proba(self, X):
        """P(y = +1 | x) under the logistic loss (``base.py:102-124``)."""
        if self.loss != "logistic":
            raise ValueError(_NO_PROBA)
        return 1 / (1 + np.exp(-self.decision_function(X)))
