"""Mirror of the ``buzhash`` package of github.com/pbs-plus/pxar v0.34.0 as the
reference uses it: ``buzhash.NewConfig(avgSize) (Config, error)`` — call sites
internal/pxarmount/commit_orchestrate.go:143-149, internal/tapeio/converter.go:248
(avg = 4 << 20) and internal/pxarmount/commit_walk_test.go:25 (avg = 4096).

The Config is a plain value (passed by value into NewPBSStore / NewLocalStore /
BackupConfig.ChunkConfig in the reference) and is consumed by ``Engine``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

WINDOW_SIZE = 64


class ConfigError(ValueError):
    """The ``error`` return of NewConfig (average size not a power of two / out of range)."""


class Config:
    __slots__ = ("_c",)

    def __init__(self, c: _lib.Config):
        self._c = c

    AvgSize = property(lambda s: int(s._c.avg))
    MinSize = property(lambda s: int(s._c.min))
    MaxSize = property(lambda s: int(s._c.max))
    WindowSize = property(lambda s: int(s._c.window))
    BreakTestMask = property(lambda s: int(s._c.mask))
    BreakTestMinimum = property(lambda s: int(s._c.break_min))

    @property
    def Table(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._c.table).copy()

    def __repr__(self) -> str:
        return (f"buzhash.Config(avg={self.AvgSize}, min={self.MinSize}, max={self.MaxSize}, "
                f"mask={self.BreakTestMask:#x}, break_min={self.BreakTestMinimum:#x})")


def NewConfig(avg_size: int, table=None) -> Config:
    """buzhash.NewConfig: derive min/max/mask from the average chunk size.

    ``table`` optionally injects the 256-word Buzhash table (default: the built-in
    casync/Proxmox table)."""
    c = _lib.Config()
    tp = None
    if table is not None:
        table = np.ascontiguousarray(table, dtype=np.uint32)
        if table.shape != (256,):
            raise ConfigError("table must hold 256 uint32 words")
        tp = table.ctypes.data
    if avg_size < 0 or avg_size >= 1 << 63:
        raise ConfigError(f"invalid average chunk size {avg_size}")
    st = _lib.lib().pbsgpu_config_init(int(avg_size), tp, C.byref(c))
    if st != _lib.OK:
        raise ConfigError(f"buzhash: average chunk size must be a power of two in [256, 2^28], got {avg_size}")
    return Config(c)


def default_table() -> np.ndarray:
    p = _lib.lib().pbsgpu_default_table()
    return np.ctypeslib.as_array(p, shape=(256,)).copy()
