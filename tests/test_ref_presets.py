"""The reference's own parameter sets as named cases (CPU part): radarays_ros_amd/params.py against the values
tests/golden/gen_refcfg.py read out of cfg/RadarModel.cfg, cfg/mulran_kaist_dyncfg*.yaml and config/*.yaml
(tests/golden/ref_presets.json), field by field.  The GPU part (each preset rendered against the oracle) is in
tests/test_gpu_round5.py."""
import dataclasses
import json
import os

import pytest

from radarays_ros_amd import params

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = json.load(open(os.path.join(GOLDEN, "ref_presets.json")))


def test_config_defaults_equal_the_cfg_file_field_by_field():
    d = params.RadarModelConfig()
    fields = {f.name for f in dataclasses.fields(d)}
    assert fields == set(REF["cfg_fields"])                         # every gen.add() and nothing else
    for name, spec in REF["cfg_fields"].items():
        v = getattr(d, name)
        assert v == spec["default"] and type(v) is {"double": float, "int": int, "bool": bool}[spec["type"]], name


@pytest.mark.parametrize("yaml_name,preset", [("mulran_kaist_dyncfg", params.kaist_preset),
                                              ("mulran_kaist_dyncfg_laserlike", params.laserlike_preset),
                                              ("mulran_kaist_dyncfg_minimal", params.minimal_preset)])
def test_presets_equal_the_reference_yaml_field_by_field(yaml_name, preset):
    want = REF["dyncfg"][yaml_name]
    got = preset()
    assert len(want) >= 30
    for name, spec in REF["cfg_fields"].items():
        # a key the file does not hold keeps the .cfg default (`dynparam load` sets only what it is given)
        v = want.get(name, spec["default"])
        assert getattr(got, name) == v, (yaml_name, name, getattr(got, name), v)
        if spec["min"] is not None:
            assert spec["min"] <= v <= spec["max"], (yaml_name, name)
    assert REF["stale_keys"][yaml_name] == ["particle_noise", "particle_noise_exp_mu"]   # not fields of the .cfg any more


def test_named_corner_values():
    k, l, m = params.kaist_preset(), params.laserlike_preset(), params.minimal_preset()
    assert k.n_samples == 50 and k.beam_width == 10.0 and k.signal_denoising_triangular_width == 35
    assert (l.n_samples, l.beam_width, l.beam_sample_dist, l.signal_denoising, l.n_reflections, l.ambient_noise) == (1, 1e-4, 0, 0, 1, 0)
    assert int(m.signal_denoising_triangular_mode * m.signal_denoising_triangular_width) == 2      # RadarCPU.cpp:57
    assert m.include_motion is True and m.signal_max == 120.0 and k.include_motion is False


def test_material_tables_equal_the_reference_yaml():
    k, o = REF["materials"]["mulran_kaist02"], REF["materials"]["oru4_test"]
    assert [list(m.astuple()) for m in params.kaist_materials()] == k["materials"]
    assert [list(m.astuple()) for m in params.oru4_test_materials()] == o["materials"]
    assert params.ORU4_OBJECT_MATERIALS == o["object_materials"] == k["object_materials"] and len(o["object_materials"]) == 18
    assert k["material_id_air"] == o["material_id_air"] == 0
    # glass is the one material of the table that transmits (0 < v < 0.3)
    assert [i for i, m in enumerate(params.oru4_test_materials()) if 0.0 < m.velocity < 0.3] == [3]
    # mulran_kaist02 lists materials 2..4 for objects its two-entry table does not have: with a one-object MulRan map
    # only object 0 (-> material 1) is ever looked up; any further object would index past the table in the reference
    assert max(k["object_materials"]) >= len(k["materials"]) and k["object_materials"][0] == 1


def test_legacy_material_tables_equal_the_reference_yaml():
    """config/oru3.yaml / oru4.yaml: the structure-of-arrays tables src/ray_reflection_test.cpp:156-167 reads -- 13 and 6 materials
    whose corner values (velocities of 0.001, BRDF exponents of 0 and 0.1) are parity cases of tests/test_gpu_round5.py."""
    for name, fn in (("oru3", params.oru3_legacy_materials), ("oru4", params.oru4_legacy_materials)):
        want = REF["materials_legacy"][name]["materials"]
        assert [list(m.astuple()) for m in fn()] == want and REF["materials_legacy"][name]["material_id_air"] == 0
    assert len(params.oru3_legacy_materials()) == 13 and len(params.oru4_legacy_materials()) == 6
    assert REF["materials_legacy"]["oru4"]["object_materials"] == params.ORU4_OBJECT_MATERIALS
