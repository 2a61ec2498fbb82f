"""CPU oracle for `vertical.pressure_on_hybrid_levels` -- TEST INFRASTRUCTURE ONLY.

NumPy restatement of /root/reference/src/earthkit/meteo/vertical/array/vertical.py:505-740
(cited below as vertical.py:N): the producer of the model-level pressure field that the thermo
hot path consumes.  Same operator order as the reference.

Parity status: PINNED by tests/golden/vertical_golden.npz (recorded from the reference by
tests/golden/gen_golden_vertical.py; includes the data of the reference's own fixture
tests/vertical/_hybrid_core_data.py) -- checked bit for bit in tests/test_vertical_oracle.py.
"""
import numpy as np

OUTPUTS = ("full", "half", "alpha", "delta")


def pressure_on_hybrid_levels(A, B, sp, levels=None, alpha_top="ifs", output="full", vertical_axis=0):
    if isinstance(output, str):  # vertical.py:623-634
        output = (output,)
    if not output:
        raise ValueError("At least one output type must be specified.")
    for out in output:
        if out not in ["full", "half", "alpha", "delta"]:
            raise ValueError(f"Unknown output type '{out}'. Allowed values are 'full', 'half', 'alpha' or 'delta'.")
    if alpha_top not in ["ifs", "arpege"]:
        raise ValueError(f"Unknown method '{alpha_top}' for pressure calculation. Use 'ifs' or 'arpege'.")

    A = np.asarray(A)
    B = np.asarray(B)
    if levels is not None:  # vertical.py:641-661: a contiguous half-level range covering the request
        nlev = A.shape[0] - 1
        levels = np.asarray(levels)
        lmax, lmin = int(levels.max()), int(levels.min())
        if lmax > nlev:
            raise ValueError(f"Requested level {lmax} exceeds the maximum number of levels {nlev}.")
        if lmin < 1:
            raise ValueError(f"Level numbering starts at 1. Found level={lmin} < 1.")
        half_idx = np.asarray(list(range(lmin - 1, lmax + 1)))
        A = A[half_idx]
        B = B[half_idx]
        out_half_idx = np.nonzero(np.asarray(levels[:, None] == half_idx[None, :]))[1]
        out_full_idx = out_half_idx - 1

    shape_half = (A.shape[0],) + (1,) * sp.ndim
    p_half = np.reshape(A, shape_half) + np.reshape(B, shape_half) * sp[np.newaxis, ...]  # vertical.py:670

    if "delta" in output or "alpha" in output:  # vertical.py:672-701
        toa = 0.1
        a_top = np.log(2) if alpha_top == "ifs" else 1.0
        shape_full = (A.shape[0] - 1,) + sp.shape
        delta = np.zeros(shape_full)
        delta[1:, ...] = np.log(p_half[2:, ...] / p_half[1:-1, ...])
        top_is_zero = np.any(p_half[0, ...] <= toa)
        if top_is_zero:
            delta[0, ...] = np.log(p_half[1, ...] / toa)
        else:
            delta[0, ...] = np.log(p_half[1, ...] / p_half[0, ...])
        alpha = np.zeros(shape_full)
        alpha[1:, ...] = 1.0 - p_half[1:-1, ...] / (p_half[2:, ...] - p_half[1:-1, ...]) * delta[1:, ...]
        if top_is_zero:
            alpha[0, ...] = a_top
        else:
            alpha[0, ...] = 1.0 - p_half[0, ...] / (p_half[1, ...] - p_half[0, ...]) * delta[0, ...]

    if "full" in output:  # vertical.py:703-710
        p_full = p_half[:-1, ...] + 0.5 * np.diff(p_half, axis=0)

    res = []
    for out in output:  # vertical.py:712-732
        if out == "full":
            res.append(p_full[out_full_idx, ...] if levels is not None else p_full)
        elif out == "half":
            res.append(p_half[out_half_idx, ...] if levels is not None else p_half)
        elif out == "alpha":
            res.append(alpha[out_full_idx, ...] if levels is not None else alpha)
        elif out == "delta":
            res.append(delta[out_full_idx, ...] if levels is not None else delta)
    if vertical_axis != 0 and res[0].ndim > 1:
        res = [np.moveaxis(r, 0, vertical_axis) for r in res]
    return res[0] if len(res) == 1 else tuple(res)


# --- geopotential chain on hybrid levels --------------------------------
# vertical.py:330-356, 472-502, 741-1190.  Constants: constants/constants.py (g, R_earth, Rd, Rv).
G0 = 9.80665
R_EARTH = 6371229
RD = 287.0597
RV = 461.51


def geopotential_height_from_geopotential(z):  # vertical.py:330-356
    return z / G0


def geometric_height_from_geopotential(z, R_earth=R_EARTH):  # vertical.py:472-502
    z = z / G0
    return R_earth * z / (R_earth - z)


def _thickness(t, q, alpha, delta):  # vertical.py:741-768
    R = RD + (RV - RD) * q  # thermo.specific_gas_constant (thermo.py:1706)
    d = R * t
    dphi_half = np.cumulative_sum(np.flip(d[1:, ...] * delta[1:, ...], axis=0), axis=0)
    dphi_half = np.flip(dphi_half, axis=0)
    dphi = np.zeros_like(d)
    dphi[:-1, ...] = dphi_half + d[:-1, ...] * alpha[:-1, ...]
    dphi[-1, ...] = d[-1, ...] * alpha[-1, ...]
    return dphi


def relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta(t, q, alpha, delta, vertical_axis=0):
    # vertical.py:815-891
    alpha, delta, t, q = (np.asarray(x) for x in (alpha, delta, t, q))
    if vertical_axis != 0:
        alpha, delta, t, q = (np.moveaxis(x, vertical_axis, 0) for x in (alpha, delta, t, q))
    dphi = _thickness(t, q, alpha, delta)
    if vertical_axis != 0:
        dphi = np.moveaxis(dphi, 0, vertical_axis)
    return dphi


def _hybrid_subset(data, A, B, vertical_axis=0):  # vertical.py:1191-1203
    nlev_t = data.shape[vertical_axis]
    nlev = A.shape[0] - 1
    if nlev_t != nlev:
        return list(range(nlev - nlev_t + 1, nlev + 1))
    return None


def relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp, alpha_top="ifs", vertical_axis=0):
    # vertical.py:894-994
    A, B, sp, t, q = (np.asarray(x) for x in (A, B, sp, t, q))
    levels = _hybrid_subset(t, A, B, vertical_axis)
    alpha, delta = pressure_on_hybrid_levels(A, B, sp, alpha_top=alpha_top, levels=levels, output=("alpha", "delta"))
    if vertical_axis != 0:
        alpha, delta, t, q = (np.moveaxis(x, vertical_axis, 0) for x in (alpha, delta, t, q))
    dphi = _thickness(t, q, alpha, delta)
    if vertical_axis != 0:
        dphi = np.moveaxis(dphi, 0, vertical_axis)
    return dphi


def geopotential_on_hybrid_levels(t, q, zs, A, B, sp, alpha_top="ifs", vertical_axis=0):  # vertical.py:997-1069
    z = relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp, vertical_axis=vertical_axis,
                                                         alpha_top=alpha_top)
    return z + np.asarray(zs)


def height_on_hybrid_levels(t, q, zs, A, B, sp, alpha_top="ifs", h_type="geometric", h_reference="ground",
                            vertical_axis=0):  # vertical.py:1072-1188
    if h_reference not in ["sea", "ground"]:
        raise ValueError(f"Unknown '{h_reference=}'. Use 'sea' or 'ground'.")
    zt = relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp, alpha_top=alpha_top,
                                                          vertical_axis=vertical_axis)
    if h_reference == "sea":
        z = zt + np.asarray(zs)
        return geometric_height_from_geopotential(z) if h_type == "geometric" else geopotential_height_from_geopotential(z)
    if h_type == "geometric":
        zs = np.asarray(zs)
        h_surf = geometric_height_from_geopotential(zs)
        return geometric_height_from_geopotential(zt + zs) - h_surf
    return geopotential_height_from_geopotential(zt)
