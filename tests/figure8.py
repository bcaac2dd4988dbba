"""The Figure-8 stand-in (tools/figure8.py: a data generator shared with bench.py --workload c3), under the name the tests import."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.figure8 import SEGS, constraints, make, raycast  # noqa: F401,E402
