from .s2st_loss import Tacotron2Criterion, label_smoothed_nll_loss  # noqa: F401
from .s2st_loss_mtl import Tacotron2MTLCriterion  # noqa: F401
from .t2s_loss import Tacotron2T2SCriterion  # noqa: F401
from .s2t_loss import LabelSmoothedCrossEntropyCriterion  # noqa: F401
