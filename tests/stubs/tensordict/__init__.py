"""Minimal stand-in for the ``tensordict`` package (NOT installed in the build or GPU images), written for the tests only: just enough of
``TensorDict`` for geometry_rl_amd.trpl's tensordict branches to execute -- construction from a dict + batch size, key access, ``get``,
``select``, ``detach``, ``keys`` / ``items``."""
import torch


class TensorDict:
    def __init__(self, source=None, batch_size=None, **kw):
        self._d = dict(source or {})
        self.batch_size = torch.Size(batch_size or [])

    def keys(self):
        return self._d.keys()

    def items(self):
        return self._d.items()

    def __contains__(self, k):
        return k in self._d

    def __getitem__(self, k):
        return self._d[k]

    def __setitem__(self, k, v):
        self._d[k] = v

    def get(self, k, default=None):
        return self._d.get(k, default)

    def set(self, k, v):
        self._d[k] = v
        return self

    def select(self, *keys, strict=True):
        if strict:
            missing = [k for k in keys if k not in self._d]
            if missing:
                raise KeyError(missing)
        return TensorDict({k: self._d[k] for k in keys if k in self._d}, self.batch_size)

    def detach(self):
        return TensorDict({k: (v.detach() if torch.is_tensor(v) else v) for k, v in self._d.items()}, self.batch_size)

    def apply(self, fn):
        return TensorDict({k: fn(v) for k, v in self._d.items()}, self.batch_size)
