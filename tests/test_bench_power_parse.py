"""bench.py's `power` block reads `rocm-smi --showclocks --showpower`: the two lines it needs, as the tool of this image prints them (ROCm 7.2)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SAMPLE = """

============================ ROCm System Management Interface ============================
=================================== Power Consumption ====================================
GPU[0]\t\t: Current Socket Graphics Package Power (W): 1383.0
==========================================================================================
============================ Current clock frequencies ============================
GPU[0]\t\t: fclk clock level: 0: (1250Mhz)
GPU[0]\t\t: mclk clock level: 0: (2000Mhz)
GPU[0]\t\t: sclk clock level: 1: (2020Mhz)
GPU[0]\t\t: socclk clock level: 3: (1143Mhz)
==========================================================================================
"""


def test_parse_smi():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_smi(SAMPLE) == (2020, 1383.0)
    assert bench.parse_smi(SAMPLE.replace("level: 1: (2020Mhz)", "level: S: (95Mhz)")) == (95, 1383.0)   # (the idle level prints as "S")
    assert bench.parse_smi("rocm-smi: no devices") is None
