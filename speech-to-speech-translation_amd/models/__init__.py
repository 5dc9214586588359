from .s2st_transformer import S2STTransformerModel, base_architecture  # noqa: F401
from .s2st_transformer_mtl import S2STTransformerMTLModel, mtl_architecture  # noqa: F401
from .t2s_transformer import T2STransformerModel, t2s_architecture  # noqa: F401
from .s2t_transformer import S2TTransformerModel, s2t_architecture  # noqa: F401
