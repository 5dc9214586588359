from .s2st_transformer import S2STTransformerModel, base_architecture  # noqa: F401
