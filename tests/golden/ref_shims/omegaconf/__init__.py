"""Throw-away stand-in for `omegaconf`, used ONLY by tests/golden/make_golden.py when it imports
the reference in the build container (omegaconf is not installed, no network).  The reference's
hot path needs just `OmegaConf.load` + attribute access/assignment (motionformer.py:93-101)."""
import yaml


class _Node(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return _Node({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            return _wrap(yaml.safe_load(f))

    @staticmethod
    def create(d):
        return _wrap(d)
