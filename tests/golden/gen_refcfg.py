#!/usr/bin/env python3
"""Reads the reference's OWN parameter sets and writes their values (numbers only) to tests/golden/ref_presets.json.

    python tests/golden/gen_refcfg.py          (in the development container: needs /root/reference)

Sources (nothing but values leaves them):
  cfg/RadarModel.cfg                              every gen.add(): name, type, default, min, max
  cfg/mulran_kaist_dyncfg{,_laserlike,_minimal}.yaml   the three dynamic-reconfigure presets (`dynparam load` files): the
                                                  top-level `dictitems` (the nested `groups` copy is dropped)
  config/mulran_kaist02.yaml, config/oru4_test.yaml    material tables: `materials`, `material_id_air`, `object_materials`
  config/oru3.yaml, config/oru4.yaml              the LEGACY structure-of-arrays tables (`velocities`, `ambient`, `diffuse`, `specular`;
                                                  read by src/ray_reflection_test.cpp:156-167 only): 13 geological / 6 office materials
                                                  with velocities down to 0.001 and fractional or zero BRDF exponents
A preset file sets only the keys it holds; what `dynparam load` leaves alone keeps the .cfg default.  Keys of a preset
that the .cfg no longer declares (particle_noise*, from an older version of the package) are listed under "stale_keys".
"""
import json
import os
import re

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_presets.json")


class Loader(yaml.SafeLoader):
    pass


def _object_new(loader, suffix, node):          # !!python/object/new:dynamic_reconfigure.encoding.Config -> plain dict
    return loader.construct_mapping(node, deep=True)


Loader.add_multi_constructor("tag:yaml.org,2002:python/object/new:", _object_new)


def cfg_fields():
    text = open(os.path.join(REF, "cfg", "RadarModel.cfg")).read()
    out = {}
    for line in text.splitlines():
        m = re.match(r'\s*gen\.add\(\s*"(\w+)"\s*,\s*(\w+)_t\s*,\s*0\s*,\s*"[^"]*"\s*,\s*([^,)]+)(?:,\s*([^,)]+)\s*,\s*([^,)]+))?', line)
        if not m:
            continue
        name, typ, default, lo, hi = m.groups()
        conv = {"double": float, "int": int, "bool": lambda s: s.strip() == "True"}[typ]
        out[name] = {"type": typ, "default": conv(default), "min": None if lo is None else conv(lo), "max": None if hi is None else conv(hi)}
    return out


def main():
    fields = cfg_fields()
    doc = {"provenance": "values read from /root/reference/cfg/*.yaml, cfg/RadarModel.cfg and config/*.yaml by tests/golden/gen_refcfg.py",
           "cfg_fields": fields, "dyncfg": {}, "stale_keys": {}, "materials": {}}
    for name in ("mulran_kaist_dyncfg", "mulran_kaist_dyncfg_laserlike", "mulran_kaist_dyncfg_minimal"):
        d = yaml.load(open(os.path.join(REF, "cfg", name + ".yaml")), Loader=Loader)["dictitems"]
        d.pop("groups", None)
        doc["stale_keys"][name] = sorted(k for k in d if k not in fields)
        doc["dyncfg"][name] = {k: d[k] for k in sorted(d) if k in fields}
    for name in ("mulran_kaist02", "oru4_test"):
        d = yaml.safe_load(open(os.path.join(REF, "config", name + ".yaml")))
        doc["materials"][name] = {
            "materials": [[m["velocity"], m["ambient"], m["diffuse"], m["specular"]] for m in d["materials"]],
            "material_id_air": d["material_id_air"], "object_materials": d["object_materials"]}
    doc["materials_legacy"] = {}
    for name in ("oru3", "oru4"):
        d = yaml.safe_load(open(os.path.join(REF, "config", name + ".yaml")))
        n = len(d["velocities"])
        assert all(len(d[k]) == n for k in ("ambient", "diffuse", "specular"))
        doc["materials_legacy"][name] = {
            "materials": [[d["velocities"][i], d["ambient"][i], d["diffuse"][i], d["specular"][i]] for i in range(n)],
            "material_id_air": d.get("material_id_air", 0), "object_materials": d.get("object_materials")}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
