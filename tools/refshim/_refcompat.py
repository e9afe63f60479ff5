"""Restores numpy/pandas APIs removed since the reference's pins (numpy 1.19 / pandas 1.1)."""
import numpy as np
import pandas as pd

for _n, _t in (("float", float), ("int", int), ("bool", bool)):
    if not hasattr(np, _n):
        setattr(np, _n, _t)

if not hasattr(pd.DataFrame, "append"):
    def _append(self, other, ignore_index=False, **kw):
        return pd.concat([self, other], ignore_index=ignore_index)
    pd.DataFrame.append = _append
