"""Fixture generator (runs in the build container only: needs /root/reference): the KEYS of the reference's configs
(fsr_vln/config/*.yaml) -- every section's key names, and for the `pipeline` section the scalar values too -- as
tests/golden/config_keys.json.  Data, not the files: tests/test_config_surface.py feeds them through Graph(cfg) / hmsg_config."""
import glob
import json
import os

import yaml

REF = "/root/reference/fsr_vln/config"
OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "config_keys.json")


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "*.yaml"))):
        cfg = yaml.safe_load(open(path)) or {}
        rec = {}
        for section, body in cfg.items():
            if not isinstance(body, dict):
                rec[section] = None
                continue
            if section == "pipeline":
                rec[section] = {k: (v if isinstance(v, (int, float, str, bool)) or v is None else str(type(v).__name__)) for k, v in body.items()}
            else:
                rec[section] = sorted(body.keys())
        out[os.path.basename(path)] = rec
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", os.path.abspath(OUT), len(out), "configs")


if __name__ == "__main__":
    main()
