"""Gabor channel constants of the homogeneous texture descriptor (scripts/const_data.py)."""
CEN_FREQ = [3 / 4.0, 3 / 8.0, 3 / 16.0, 3 / 32.0, 3 / 64.0]
OCT_BAND = [1 / 2.0, 1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0]
ANGULAR_BAND = [30, 30, 30, 30, 30, 30]
ANGULAR_CEN = [0, 30, 60, 90, 120, 150]
