"""Number formatting that matches what Julia's print/writedlm emit for Float64, so that the
reference's consumers (scripts/aggregate_mcmc.jl:71 `eval(Meta.parse(rhs))`, scripts/compare_mcmc.jl)
read our files unchanged.  Julia prints the shortest round-trip decimal; fixed notation when
1e-4 <= |x| < 1e6 (always with a fractional part), otherwise `d.ddde±x` with at least one fractional
digit and no padding of the exponent."""
from __future__ import annotations

import math


def jl_float(x: float) -> str:
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    r = repr(x)  # shortest round-trip digits, like Ryu
    mant, _, exp = r.partition("e")
    sign = "-" if mant.startswith("-") else ""
    mant = mant.lstrip("-")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    # decimal exponent of the first significant digit
    if exp:
        e10 = int(exp) + len(ip) - 1
    else:
        if ip.strip("0"):
            e10 = len(ip.lstrip("0")) - 1
        else:
            e10 = -(len(fp) - len(fp.lstrip("0")) + 1)
    digits = digits.rstrip("0") or "0"
    if -5 < e10 < 6:
        if e10 >= 0:
            whole = digits[: e10 + 1].ljust(e10 + 1, "0")
            frac = digits[e10 + 1:] or "0"
        else:
            whole = "0"
            frac = "0" * (-e10 - 1) + digits
        return f"{sign}{whole}.{frac}"
    frac = digits[1:] or "0"
    return f"{sign}{digits[0]}.{frac}e{e10}"


def jl_vector(v) -> str:
    """`[a, b, c]` -- how string interpolation of a Vector{Float64} prints (mcmc_eap_chain.jl:386)."""
    return "[" + ", ".join(jl_float(x) for x in v) + "]"


def jl_row(values) -> str:
    """One writedlm(io, row, ',') line: hcat promotes everything (also `step`) to Float64."""
    return ",".join(jl_float(x) for x in values)
