"""Regenerates the committed fixtures under tests/golden/ (run in the BUILD container only;
/root/reference does not exist on the GPU box and no test reads it).

What is copied here is DATA the reference holds for this path, never source text:
  * examples/getting_started/arrhenius_params.bson -> arrhenius_params.json (the 30 Ea / A
    Float64 values, decoded from BSON.jl's raw little-endian array encoding);
  * the known-answer values asserted in test/Main/conditions.jl -> conditions_kat.json
    (numbers transcribed from the @test lines cited per entry);
  * the 5-species CRN of docs/src/tutorials/ode-solution.md:23-41 -> doc_crn.json
    (topology + the ODE right-hand sides written out in the doc, as coefficient lists).
The reference itself cannot be executed (Julia; no toolchain), so there are no
reference-produced output vectors: parity of the solve is UNPINNED.

High-accuracy truth trajectories (SciPy Radau, rtol 1e-12) for the small known-answer
networks are produced by make_truth.py next to this file.
"""
import json
import os

import bson
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    raw = open(os.path.join(REF, "examples/getting_started/arrhenius_params.bson"), "rb").read()
    d = bson.decode(raw)
    out = {}
    for key in ("Ea", "A"):
        assert d[key]["tag"] == "array" and d[key]["type"]["name"] == ["Core", "Float64"]
        arr = np.frombuffer(d[key]["data"], dtype="<f8")
        assert list(d[key]["size"]) == [len(arr)]
        out[key] = [float(x) for x in arr]
    out["source"] = "examples/getting_started/arrhenius_params.bson"
    json.dump(out, open(os.path.join(HERE, "arrhenius_params.json"), "w"), indent=1)

    kat = {
        "source": "test/Main/conditions.jl",
        "lineardirect": {"args": {"rate": 50.0, "X_start": 300.0, "X_end": 500.0},
                         "t_end": 4.0, "f_at": [[2.0, 400.0]], "tstops": [4.0], "lines": "17-27"},
        "nulldirect": {"args": {"X_start": 300.0, "t_end": 10.0}, "f_at": [[5.0, 300.0]], "tstops": [10.0],
                       "lines": "8-15"},
        "nullgradient": {"args": {"X_start": 300.0, "t_end": 10.0}, "grad_at": [[5.0, 0.0]], "tstops": [10.0],
                         "lines": "29-36"},
        "lineargradient": {"args": {"rate": 50.0, "X_start": 300.0, "X_end": 500.0}, "t_end": 4.0,
                           "grad_at": [[2.0, 50.0], [5.0, 0.0]], "tstops": [4.0], "lines": "38-49"},
        "doubleramp": {"args": {"X_start": 300.0, "t_start_plateau": 5.0, "rate1": 10.0, "X_mid": 500.0,
                                "t_mid_plateau": 3.0, "rate2": -20.0, "X_end": 200.0, "t_end_plateau": 5.0},
                       "t_end": 48.0, "t_blend": 0.0, "tstops": [5.0, 25.0, 28.0, 43.0, 48.0],
                       "grad_at": [[1.0, 0.0], [15.0, 10.0], [27.0, 0.0], [35.0, -20.0], [45.0, 0.0], [100.0, 0.0]],
                       "lines": "51-76"},
        "doubleramp_blended": {"args": {"X_start": 300.0, "t_start_plateau": 5.0, "rate1": 10.0, "X_mid": 500.0,
                                        "t_mid_plateau": 3.0, "rate2": -20.0, "X_end": 200.0, "t_end_plateau": 5.0,
                                        "t_blend": 0.1},
                               "tstops": [4.9, 5.1, 24.9, 25.1, 27.9, 28.1, 42.9, 43.1, 48.0], "lines": "78-89"},
    }
    json.dump(kat, open(os.path.join(HERE, "conditions_kat.json"), "w"), indent=1)

    # docs/src/tutorials/ode-solution.md:23-41: A <-> B + C (k1, k-1); B <-> D (k2, k-2); C + D <-> E (k3, k-3)
    # species order A,B,C,D,E ; reaction order k1, k-1, k2, k-2, k3, k-3
    doc = {
        "source": "docs/src/tutorials/ode-solution.md:23-41",
        "species": ["A", "B", "C", "D", "E"],
        "reacs": [[[0, 1]], [[1, 1], [2, 1]], [[1, 1]], [[3, 1]], [[2, 1], [3, 1]], [[4, 1]]],
        "prods": [[[1, 1], [2, 1]], [[0, 1]], [[3, 1]], [[1, 1]], [[4, 1]], [[2, 1], [3, 1]]],
        # each ODE as a list of [sign, k index, [species factors]] exactly as printed in the doc
        "odes": {
            "A": [[-1, 0, [0]], [1, 1, [1, 2]]],
            "B": [[1, 0, [0]], [1, 3, [3]], [-1, 2, [1]], [-1, 1, [1, 2]]],
            "C": [[1, 0, [0]], [1, 5, [4]], [-1, 1, [1, 2]], [-1, 4, [2, 3]]],
            "D": [[1, 2, [1]], [1, 5, [4]], [-1, 3, [3]], [-1, 4, [2, 3]]],
            "E": [[-1, 5, [4]], [1, 4, [2, 3]]],
        },
    }
    json.dump(doc, open(os.path.join(HERE, "doc_crn.json"), "w"), indent=1)
    print("wrote arrhenius_params.json, conditions_kat.json, doc_crn.json")


if __name__ == "__main__":
    main()
