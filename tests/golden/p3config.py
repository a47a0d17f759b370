"""Materialises primer3's parameter tables (fixture data in primer3_params.json) as the *.ds / *.dh files thal expects."""
import json
import os
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_dir = None


def config_dir() -> str:
    global _dir
    if _dir is None:
        d = tempfile.mkdtemp(prefix="primer3_config_")
        tabs = json.load(open(os.path.join(_HERE, "primer3_params.json")))["tables"]
        for name, rows in tabs.items():
            with open(os.path.join(d, name), "w") as f:
                for r in rows:
                    f.write("\t".join(r) + "\n")
        _dir = d + "/"
    return _dir
