#!/usr/bin/env python3
"""PMC calibration target: a float4 copy of exactly 1 GiB read + 1 GiB written per launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lbm_amd
print("copy GB/s:", lbm_amd.copy_bandwidth_gbps(1 << 30, 4))
