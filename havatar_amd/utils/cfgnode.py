"""Minimal attribute-access config node (role of the reference's YACS-style utils/cfgnode.py: cfg.nerf.validation.num_coarse,
getattr(cfg.nerf, mode).chunksize, ...).  Nested dicts become nested CfgNodes; values stay mutable."""
import yaml


class CfgNode(dict):
    def __init__(self, init_dict=None):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    @classmethod
    def load_yaml(cls, path):
        with open(path, "r") as f:
            return cls(yaml.load(f, Loader=yaml.FullLoader))

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kwargs):
        """YAML text of the node (reference utils/cfgnode.py:167-187), written next to the logs by the training harness."""
        return yaml.safe_dump(self.to_dict(), **kwargs)
