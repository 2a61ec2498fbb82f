"""(function, argument names, kwargs) for every public thermo function and variant -- shared by
the golden generator and the GPU tests."""

PHASES = ["mixed", "water", "ice"]
EPT = ["ifs", "bolton35", "bolton39"]


def case_table():
    """(function, argument names, kwargs) for every public function and variant."""
    c = []
    add = lambda f, a, **k: c.append((f, a, k))  # noqa: E731
    add("celsius_to_kelvin", ["tc"])
    add("kelvin_to_celsius", ["t"])
    add("specific_humidity_from_mixing_ratio", ["w"])
    add("mixing_ratio_from_specific_humidity", ["q"])
    add("vapour_pressure_from_specific_humidity", ["q", "p"])
    add("vapour_pressure_from_mixing_ratio", ["w", "p"])
    add("specific_humidity_from_vapour_pressure", ["e", "p"])
    add("specific_humidity_from_vapour_pressure", ["es", "p2"], eps=5.0e4)
    add("mixing_ratio_from_vapour_pressure", ["e", "p"])
    add("mixing_ratio_from_vapour_pressure", ["es", "p2"], eps=5.0e4)
    for ph in PHASES:
        add("saturation_vapour_pressure", ["t"], phase=ph)
        add("saturation_mixing_ratio", ["t", "p"], phase=ph)
        add("saturation_specific_humidity", ["t", "p"], phase=ph)
        add("saturation_vapour_pressure_slope", ["t"], phase=ph)
        add("saturation_mixing_ratio_slope", ["t", "p"], phase=ph)
        add("saturation_specific_humidity_slope", ["t", "p"], phase=ph)
    add("saturation_mixing_ratio_slope", ["t", "p2"], eps=5.0e4)
    add("saturation_specific_humidity_slope", ["t", "p2"], eps=5.0e4)
    add("temperature_from_saturation_vapour_pressure", ["es"])
    add("relative_humidity_from_dewpoint", ["t", "td"])
    add("relative_humidity_from_specific_humidity", ["t", "q", "p"])
    add("specific_humidity_from_dewpoint", ["td", "p"])
    add("mixing_ratio_from_dewpoint", ["td", "p"])
    add("specific_humidity_from_relative_humidity", ["t", "r", "p"])
    add("dewpoint_from_relative_humidity", ["t", "r"])
    add("dewpoint_from_specific_humidity", ["q", "p"])
    add("virtual_temperature", ["t", "q"])
    add("virtual_potential_temperature", ["t", "q", "p"])
    add("potential_temperature", ["t", "p"])
    add("temperature_from_potential_temperature", ["th", "p"])
    add("pressure_on_dry_adiabat", ["t2", "t", "p"])
    add("temperature_on_dry_adiabat", ["p2", "t", "p"])
    for m in ("davies", "bolton"):
        add("lcl_temperature", ["t", "td"], method=m)
        add("lcl", ["t", "td", "p"], method=m)
    for m in EPT:
        add("ept_from_dewpoint", ["t", "td", "p"], method=m)
        add("ept_from_specific_humidity", ["t", "q", "p"], method=m)
        add("saturation_ept", ["t", "p"], method=m)
        for tm in ("bisect", "newton"):
            add("temperature_on_moist_adiabat", ["ept", "p"], ept_method=m, t_method=tm)
            add("wet_bulb_temperature_from_dewpoint", ["t", "td", "p"], ept_method=m, t_method=tm)
            add("wet_bulb_temperature_from_specific_humidity", ["t", "q", "p"], ept_method=m, t_method=tm)
        for tm in ("direct", "bisect", "newton"):
            add("wet_bulb_potential_temperature_from_dewpoint", ["t", "td", "p"], ept_method=m, t_method=tm)
            add("wet_bulb_potential_temperature_from_specific_humidity", ["t", "q", "p"], ept_method=m, t_method=tm)
    add("specific_gas_constant", ["q"])
    return c
