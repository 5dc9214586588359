from .s2s_translation import S2ST_TranslationTask, Dictionary  # noqa: F401
from .s2s_translation_mtl import S2ST_TranslationMTLTask  # noqa: F401
