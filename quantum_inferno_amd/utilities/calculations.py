"""Point-count rounding used to size the STFT segment (mirror of the two functions of
quantum_inferno/utilities/calculations.py that sit on the TFR path, :160-205)."""
import numpy as np

ROUNDING_TYPES = ["floor", "ceil", "round", "ceil_power_of_two", "floor_power_of_two"]
OUTPUT_TYPES = ["points", "log2", "pow2"]


def round_value(value: float, rounding_type: str = "round") -> int:
    """Round to an int by the named rule; "round" is half-to-even (ref calculations.py:160-184)."""
    if rounding_type not in ROUNDING_TYPES:
        raise ValueError(f"Invalid rounding type {rounding_type}, must be one of {ROUNDING_TYPES}")
    if rounding_type == "floor":
        return int(np.floor(value))
    if rounding_type == "ceil":
        return int(np.ceil(value))
    if rounding_type == "round":
        return int(np.round(value))
    exponent = np.ceil(np.log2(value)) if rounding_type == "ceil_power_of_two" else np.floor(np.log2(value))
    return 2 ** int(exponent)


def get_num_points(sample_rate_hz: float, duration_s: float, rounding_type: str, output_unit: str) -> int:
    """Points (or their log2 / pow2) in duration_s at sample_rate_hz (ref calculations.py:187-205)."""
    if output_unit not in OUTPUT_TYPES:
        raise ValueError(f"Invalid output unit {output_unit}, must be one of {OUTPUT_TYPES}")
    points = sample_rate_hz * duration_s
    if output_unit == "log2":
        points = np.log2(points)
    elif output_unit == "pow2":
        points = 2 ** points
    return round_value(points, rounding_type)
