#!/usr/bin/env python3
"""The index-checked build reports: with a library built with -DAMVS_CHECK_INDICES -DAMVS_CHECK_SELFTEST (one
deliberately out-of-range index per fast sweep launch, never dereferenced) a sweep must return AMVS_EINDEX and
amvs_index_check must name translation unit 2 (amvs_kernels_fast.hip).

    ALL=1 tools/build_variant.sh checkself -DAMVS_CHECK_INDICES -DAMVS_CHECK_SELFTEST
    AMVS_LIB=$PWD/build/variants/libamvs_checkself.so python tools/index_check_selftest.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    import amvs
    from amvs import _lib
    from amvs.engine import make_pm_params
    from conftest import GoldenScene
    assert _lib.index_checks_enabled(), "not an index-checked build: " + _lib.load().amvs_version().decode()
    sc = GoldenScene("scene_a")
    with sc.engine("fast") as eng:
        p = make_pm_params(7, 1, 1, sc.depth_min, sc.depth_max)
        try:
            eng.patchmatch([2], [[1, 3, 0, 4]], p, 1)
        except amvs.AmvsError as e:
            print("sweep refused as expected:", e)
        else:
            raise SystemExit("the deliberate violation was NOT reported")
    count, tu, line, index, extent = _lib.index_check(reset=True)
    print(f"report: {count} violations, first in translation unit {tu} line {line}: index {index}, extent {extent}")
    assert count >= 1 and tu == 2 and index == extent + 7
    assert _lib.index_check()[0] == 0, "the report was not reset"
    print("index-check self test ok")


if __name__ == "__main__":
    main()
