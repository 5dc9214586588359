from .s2s_translation import S2ST_TranslationTask, Dictionary  # noqa: F401
