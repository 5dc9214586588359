class HydraConfig:
    @staticmethod
    def get():
        raise RuntimeError("hydra stub")
