"""Saving results (``flooder/io.py:14-57``): ``torch.save`` with an optional ``_meta`` record for dictionaries."""
from __future__ import annotations

import datetime
from pathlib import Path
from typing import Any, Union

import torch


def save_to_disk(obj: Any, path: Union[str, Path], metadata: bool = True, overwrite: bool = False) -> None:
    """Write ``obj`` with ``torch.save``.  A dict gets (on a copy, and only if it has none) a ``"_meta"`` entry
    ``{"timestamp": iso time, "keys": list of its keys}`` when ``metadata`` is true.  An existing file is kept
    and ``FileExistsError`` raised unless ``overwrite`` is true."""
    target = Path(path)
    if target.exists() and not overwrite:
        raise FileExistsError(f"File already exists: {target}")
    payload = obj
    if metadata and isinstance(obj, dict):
        payload = dict(obj)
        payload.setdefault("_meta", {"timestamp": datetime.datetime.now().isoformat(), "keys": list(obj.keys())})
    torch.save(payload, target)
