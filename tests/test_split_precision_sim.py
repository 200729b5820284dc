"""CPU check of the precision argument behind the two-piece fp16 split (flavour b): on emulated pieces,
the recurrence's error against float64 is that of a genuine fp32 evaluation."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_piece_fp16_products_are_as_accurate_as_fp32_products():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "split_precision_sim.py"), "1", "60"],
                         capture_output=True, text=True, timeout=600, check=True).stdout
    rms = {m.group(1): float(m.group(2)) for m in re.finditer(r"^(\S+)\s+max err \S+\s+rms err (\S+)$", out, re.M)}
    assert {"f32", "bf16x3", "f16x2_3", "f16x2_4"} <= set(rms), out
    assert rms["f16x2_3"] <= 1.25 * rms["f32"], rms          # measured 1.03x
    assert rms["f16x2_4"] <= 1.25 * rms["f32"], rms
    assert rms["bf16x3"] <= rms["f32"], rms                  # three bf16 pieces: exact products, fewer roundings
