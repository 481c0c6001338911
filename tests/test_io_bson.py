"""CPU tests of save_output / load_output (reference src/analysis/io.jl:70-255).

PINNED against the reference: the one BSON.jl file it ships, examples/getting_started/arrhenius_params.bson - committed
here as the fixture tests/golden/arrhenius_params.bson.b64 (data, base64 of the 758 bytes) - is reproduced byte for byte
by decoding it and re-encoding the decoded arrays with this module's BSON.jl lowering. Everything that file does not
exercise (Symbols, tuples, arrays of arrays, non-Symbol dictionaries, VersionNumber) follows BSON.jl's published rules
and is only checked for self-consistency (round trip, valid BSON for an independent decoder, the reference's key tree)."""
import base64
import os

import numpy as np
import pytest

from kinetica_jl_amd import conditions as C
from kinetica_jl_amd import io as kio
from kinetica_jl_amd import solving as S


def test_float64_array_encoding_matches_the_reference_file_byte_for_byte(golden_dir):
    raw = base64.b64decode(open(os.path.join(golden_dir, "arrhenius_params.bson.b64")).read())
    assert len(raw) == 758
    tree = kio.loads(raw)
    assert list(tree) == ["Ea", "A"] and tree["Ea"]["tag"] == "array" and tree["Ea"]["type"]["name"] == ["Core", "Float64"]
    vals = kio.raise_(tree)
    assert vals["Ea"].dtype == np.float64 and vals["Ea"].shape == (30,) and vals["A"].shape == (30,)
    assert kio.dumps(kio.lower(vals)) == raw
    # the same numbers as the JSON transcription the Arrhenius tests use
    import json
    j = json.load(open(os.path.join(golden_dir, "arrhenius_params.json")))
    np.testing.assert_array_equal(vals["Ea"], np.array(j["Ea"]))
    np.testing.assert_array_equal(vals["A"], np.array(j["A"]))
    # an independent BSON decoder reads what this module writes
    bson = pytest.importorskip("bson")
    d = bson.decode(kio.dumps(kio.lower(vals)))
    assert d["Ea"]["size"] == [30] and len(d["Ea"]["data"]) == 240


def make_output():
    sd = S.SpeciesData.from_names(["C", "[CH3]", "[H]", "CC"])
    rd = S.RxData(2, [[1], [2, 2]], [[2, 3], [4]], [[1], [2]], [[1, 1], [1]], dH=[4.5, -3.25])
    pars = S.ODESimulationParams(tspan=(0.0, 2.0), u0={"C": 1.0}, solve_chunkstep=0.5, save_interval=0.25)
    cs = C.ConditionSet({"T": C.LinearGradientProfile(rate=100.0, X_start=500.0, X_end=700.0), "V": 2.5}, ts_update=0.5)
    C.solve_variable_conditions(cs, pars)
    rng = np.random.default_rng(3)
    t = np.arange(9) * 0.25
    u = rng.random((9, 4))
    tst = C.get_tstops(cs)
    sol_k = S.DiscreteRates(tst, rng.random((len(tst), 2)))
    sol = S.ODESolution(t, u, "Success", k=sol_k)
    return S.ODESolveOutput(sd, rd, sol, sol_k, None, pars, cs)


def test_save_load_round_trip(tmp_path):
    out = make_output()
    f = str(tmp_path / "out.bson")
    kio.save_output(out, f)
    back = kio.load_output(f)
    assert back.sd.toInt == out.sd.toInt and back.sd.n == 4 and back.sd.toStr[2] == "[CH3]"
    assert back.rd.nr == 2 and back.rd.id_reacs == out.rd.id_reacs and back.rd.stoic_prods == out.rd.stoic_prods
    assert back.rd.dH == out.rd.dH
    np.testing.assert_array_equal(back.sol.t, out.sol.t)
    np.testing.assert_array_equal(back.sol.u, out.sol.u)
    np.testing.assert_array_equal(back.sol_k.u, out.sol_k.u)
    np.testing.assert_array_equal(back.sol_k.t, out.sol_k.t)
    assert back.pars.tspan == (0.0, 2.0) and back.pars.u0 == {"C": 1.0} and back.pars.save_interval == 0.25
    assert back.pars.low_k_cutoff == "auto" and back.pars.solve_chunkstep == 0.5 and back.pars.maxiters == 100000
    assert back.conditions.symbols == out.conditions.symbols and back.conditions.discrete_updates
    assert back.conditions.ts_update == 0.5
    pT, pV = back.conditions.profiles
    assert isinstance(pT, C.LinearGradientProfile) and pT.rate == 100.0 and pT.X_end == 700.0
    np.testing.assert_array_equal(pT.sol.u, out.conditions.profiles[0].sol.u)
    np.testing.assert_array_equal(pT.tstops, out.conditions.profiles[0].tstops)
    with pytest.raises(RuntimeError):
        pT.grad(0.0, pT)                                   # loaded_profile_null_func (io.jl:258-260)
    assert isinstance(pV, C.StaticConditionProfile) and pV.value == 2.5
    # res.sol(t): linear interpolation still works on the reloaded solution
    np.testing.assert_allclose(back.sol(0.125)[0], 0.5 * (out.sol.u[0] + out.sol.u[1]))


def test_tree_has_the_references_keys_and_is_valid_bson(tmp_path):
    bson = pytest.importorskip("bson")
    f = str(tmp_path / "out.bson")
    kio.save_output(make_output(), f)
    d = bson.decode(open(f, "rb").read())
    assert list(d) == ["KineticaCoreVersion", "sd", "rd", "pars", "sol", "conditions"]          # io.jl:108-155
    assert list(d["sd"]) == ["toInt", "n", "xyz", "level_found"]
    assert list(d["rd"]) == ["nr", "mapped_rxns", "id_reacs", "id_prods", "stoic_reacs", "stoic_prods", "dH", "rhash", "level_found"]
    assert list(d["pars"]) == ["tspan", "u0", "solver", "jac", "sparse", "adaptive_tols", "update_tols", "solve_chunks",
                               "solve_chunkstep", "maxiters", "ban_negatives", "progress", "save_interval", "low_k_cutoff",
                               "allow_short_u0"]
    assert list(d["sol"]) == ["u", "t", "vcs", "k"] and list(d["conditions"]) == ["symbols", "profiles", "discrete_updates", "ts_update"]
    assert d["sol"]["u"]["tag"] == "array" and d["sol"]["u"]["size"] == [9] and d["sol"]["u"]["data"][0]["size"] == [4]
    assert d["pars"]["tspan"] == {"tag": "tuple", "data": [0.0, 2.0]}
    assert d["pars"]["low_k_cutoff"] == {"tag": "symbol", "name": "auto"}
    # Vector{Symbol} / the keys and values of Dict{String,Int}: TYPED vectors, i.e. tagged `array` documents (BSON.jl lowers
    # every Array except Vector{Any} that way; a plain BSON array would load as Vector{Any} in Julia)
    sy = d["conditions"]["symbols"]
    assert sy["tag"] == "array" and sy["type"]["name"] == ["Core", "Symbol"] and sy["size"] == [2]
    assert sy["data"][0] == {"tag": "symbol", "name": "T"}
    keys, vals = d["sd"]["toInt"]["data"]
    assert keys["tag"] == "array" and keys["type"]["name"] == ["Core", "String"] and isinstance(keys["data"], list)
    assert vals["tag"] == "array" and vals["type"]["name"] == ["Core", "Int64"] and isinstance(vals["data"], bytes)
    assert d["sol"]["vcs"] is None and d["sol"]["k"]["t"]["size"] == [5]


def test_profile_type_names_are_looked_up_in_a_fixed_list(tmp_path):
    """The reference evaluates the profile type's name (io.jl:238); here a file is data: an unknown name is an error."""
    f = str(tmp_path / "out.bson")
    kio.save_output(make_output(), f)
    tree = kio.loads(open(f, "rb").read())
    tree["conditions"]["profiles"][0]["pType"]["name"] = "ConditionSet"
    open(f, "wb").write(kio.dumps(tree))
    with pytest.raises(ValueError, match="unknown condition profile type"):
        kio.load_output(f)
