"""effective clock (GRBM_GUI_ACTIVE / 8 / launch time) per kernel from one rocprofv3 --pmc GRBM_GUI_ACTIVE pass: python tools/clk_of.py DIR"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import _demangled, read  # noqa: E402

rows = _demangled(read(sys.argv[1], "GRBM_GUI_ACTIVE"))
out = sorted(((us / max(n, 1), name, grid, val / 8.0 / (us * 1000.0)) for (name, grid), (val, n, us) in rows.items() if us > 0), reverse=True)
for us, name, grid, ghz in out[:6]:
    print("%8.1f us  %.3f GHz  %s" % (us, ghz, name[:110]))
