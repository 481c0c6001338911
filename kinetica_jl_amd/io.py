"""`save_output` / `load_output` of ODESolveOutput (reference: src/analysis/io.jl:70-158, 171-255): the BSON file the
exploration loop checkpoints per level (src/exploration/methods.jl:223-228), as a package-independent dictionary tree.

The reference writes it with BSON.jl, an un-vendored dependency, whose "lowering" of Julia values onto BSON is what a
reader has to reproduce. What can be PINNED here is pinned byte for byte: the reference ships one BSON.jl file,
examples/getting_started/arrhenius_params.bson (two Vector{Float64}), and `dumps(loads(file))` reproduces it exactly
(tests/test_io_bson.py) - document framing, key order, Int64 sizes, the `array` / `datatype` documents and the raw
little-endian `data` payload. Everything else follows BSON.jl's published lowering rules as restated below and is
UNVERIFIABLE in this image (no Julia): Symbols, tuples, arrays of arrays, dictionaries with non-Symbol keys,
VersionNumber. Those encodings are isolated in `lower_*` helpers so that a maintainer with a Julia toolchain can check
them against `BSON.load` in minutes; the tree itself (keys and nesting of `savedict`, io.jl:108-155) is the reference's.

BSON.jl lowering rules used (BSON.jl/src/extensions.jl, write.jl):
  Nothing, Bool, Int32, Int64, Float64, String, Vector{UInt8}: BSON primitives
  Dict{Symbol,Any}            -> document (keys written as they are)
  Vector{Any}                 -> BSON array
  Symbol s                    -> {tag: "symbol", name: String(s)}
  Tuple t                     -> {tag: "tuple", data: [t...]}
  DataType T                  -> {tag: "datatype", params: [T.parameters...], name: [module path..., name]}
  Array{T,N} a                -> {tag: "array", type: T, size: [size(a)...], data: isbitstype(T) ? raw bytes : [a...]}
  other structs (Dict{K,V}, VersionNumber, OrderedDict) -> {tag: "struct", type: T, data: [fields / keys, values]}
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np


# ---- plain BSON (bsonspec.org 1.1): the subset BSON.jl emits ---------------------------------------------------
class Int32(int):
    """Marks an integer that must be written as BSON int32 (Julia Int32)."""


def _cstr(s):
    b = s.encode("utf-8")
    assert b"\x00" not in b
    return b + b"\x00"


def _enc_value(v):
    """(type byte, payload)"""
    if v is None:
        return b"\x0a", b""
    if isinstance(v, (bool, np.bool_)):
        return b"\x08", b"\x01" if v else b"\x00"
    if isinstance(v, Int32):
        return b"\x10", struct.pack("<i", int(v))
    if isinstance(v, (int, np.integer)):
        return b"\x12", struct.pack("<q", int(v))
    if isinstance(v, (float, np.floating)):
        return b"\x01", struct.pack("<d", float(v))
    if isinstance(v, str):
        b = v.encode("utf-8") + b"\x00"
        return b"\x02", struct.pack("<i", len(b)) + b
    if isinstance(v, (bytes, bytearray)):
        return b"\x05", struct.pack("<i", len(v)) + b"\x00" + bytes(v)
    if isinstance(v, dict):
        return b"\x03", dumps(v)
    if isinstance(v, (list, tuple)):
        return b"\x04", dumps(OrderedDict((str(i), x) for i, x in enumerate(v)))
    raise TypeError(f"cannot encode {type(v).__name__} as BSON")


def dumps(doc) -> bytes:
    """A document (dict with str keys, insertion order kept) as BSON bytes."""
    body = b""
    for k, v in doc.items():
        t, p = _enc_value(v)
        body += t + _cstr(str(k)) + p
    return struct.pack("<i", len(body) + 5) + body + b"\x00"


def _dec_doc(b, pos, as_list):
    size = struct.unpack_from("<i", b, pos)[0]
    end = pos + size - 1
    pos += 4
    out = [] if as_list else OrderedDict()
    while pos < end:
        t = b[pos]
        pos += 1
        z = b.index(b"\x00", pos)
        key = b[pos:z].decode("utf-8")
        pos = z + 1
        if t == 0x01:
            v = struct.unpack_from("<d", b, pos)[0]; pos += 8
        elif t == 0x02:
            n = struct.unpack_from("<i", b, pos)[0]
            v = b[pos + 4:pos + 4 + n - 1].decode("utf-8"); pos += 4 + n
        elif t in (0x03, 0x04):
            v, pos = _dec_doc(b, pos, t == 0x04)
        elif t == 0x05:
            n = struct.unpack_from("<i", b, pos)[0]
            v = bytes(b[pos + 5:pos + 5 + n]); pos += 5 + n
        elif t == 0x08:
            v = b[pos] != 0; pos += 1
        elif t == 0x0a:
            v = None
        elif t == 0x10:
            v = Int32(struct.unpack_from("<i", b, pos)[0]); pos += 4
        elif t == 0x12:
            v = struct.unpack_from("<q", b, pos)[0]; pos += 8
        else:
            raise ValueError(f"unsupported BSON element type 0x{t:02x}")
        if as_list:
            out.append(v)
        else:
            out[key] = v
    assert b[end] == 0
    return out, end + 1


def loads(b: bytes):
    doc, pos = _dec_doc(b, 0, False)
    assert pos == len(b)
    return doc


# ---- BSON.jl lowering ---------------------------------------------------------------------------------------------
_JL_ELTYPE = {np.dtype("float64"): ["Core", "Float64"], np.dtype("int64"): ["Core", "Int64"],
              np.dtype("int32"): ["Core", "Int32"], np.dtype("uint8"): ["Core", "UInt8"], np.dtype("bool"): ["Core", "Bool"]}


def lower_datatype(name, params=()):
    """DataType -> {tag, params, name} in the key order BSON.jl writes (as in the reference's own file)."""
    return OrderedDict([("tag", "datatype"), ("params", list(params)), ("name", list(name))])


def lower_bits_array(a):
    """Array of a bits type: raw little-endian column-major payload. PINNED by arrhenius_params.bson."""
    a = np.asarray(a)
    flat = np.asfortranarray(a).reshape(-1, order="F").astype(a.dtype.newbyteorder("<"), copy=False)
    return OrderedDict([("tag", "array"), ("type", lower_datatype(_JL_ELTYPE[a.dtype])), ("size", [int(s) for s in a.shape]),
                        ("data", flat.tobytes())])


def lower_symbol(s):
    return OrderedDict([("tag", "symbol"), ("name", str(s))])


def lower_tuple(t):
    return OrderedDict([("tag", "tuple"), ("data", [lower(x) for x in t])])


def _vector_datatype(eltype_doc):
    return lower_datatype(["Core", "Array"], [eltype_doc, 1])


def lower_vector_of_vectors(rows, dtype):
    """Vector{Vector{T}} (sol.u, sol_k.u, rd.id_reacs ...): an `array` document whose data is a BSON array of lowered
    inner arrays."""
    return OrderedDict([("tag", "array"), ("type", _vector_datatype(lower_datatype(_JL_ELTYPE[np.dtype(dtype)]))),
                        ("size", [len(rows)]), ("data", [lower_bits_array(np.asarray(r, dtype=dtype)) for r in rows])])


def lower_typed_vector(items, eltype_name):
    """Vector{T} of a NON-bits element type other than Any (Vector{String}, Vector{Symbol}): BSON.jl lowers every Array
    except Vector{Any} to a tagged `array` document {tag, type = eltype, size, data}; for non-bits elements `data` is the
    BSON array of the lowered elements (BSON.jl's `lower(x::Array)`; a plain BSON array would come back as Vector{Any})."""
    return OrderedDict([("tag", "array"), ("type", lower_datatype(eltype_name)), ("size", [len(items)]),
                        ("data", [lower(v) for v in items])])


def _lower_vector_of(values, dt):
    name = dt["name"]
    if name == ["Core", "String"]:
        return lower_typed_vector([str(v) for v in values], name)
    if name == ["Core", "Int64"]:
        return lower_bits_array(np.asarray(list(values), dtype=np.int64))
    if name == ["Core", "Float64"]:
        return lower_bits_array(np.asarray(list(values), dtype=np.float64))
    return lower(list(values))


def lower_dict(d, ktype, vtype):
    """Dict{K,V} with non-Symbol keys: BSON.jl's struct form, data = [keys, values] with keys :: Vector{K} and
    values :: Vector{V} lowered as TYPED vectors."""
    ks, vs = list(d.keys()), list(d.values())
    return OrderedDict([("tag", "struct"), ("type", lower_datatype(["Base", "Dict"], [ktype, vtype])),
                        ("data", [_lower_vector_of(ks, ktype), _lower_vector_of(vs, vtype)])])


class Symbol(str):
    """A Julia Symbol value (distinct from a String when lowered)."""


def lower(x):
    """Python stand-ins -> BSON.jl documents: Symbol, tuple, numpy arrays, lists (Vector{Any}), dicts with str keys
    (Dict{Symbol,Any}); scalars pass through."""
    if isinstance(x, Symbol):
        return lower_symbol(x)
    if isinstance(x, tuple):
        return lower_tuple(x)
    if isinstance(x, np.ndarray):
        return lower_bits_array(x)
    if isinstance(x, list):
        return [lower(v) for v in x]
    if isinstance(x, dict):
        return OrderedDict((str(k), lower(v)) for k, v in x.items())
    return x


def raise_(doc):
    """The inverse of `lower` for the documents this module writes (and for arrhenius_params.bson)."""
    if isinstance(doc, list):
        return [raise_(v) for v in doc]
    if not isinstance(doc, dict):
        return doc
    tag = doc.get("tag")
    if tag == "array":
        name = doc["type"]["name"]
        size = [int(s) for s in doc["size"]]
        if isinstance(doc["data"], (bytes, bytearray)):
            dt = {tuple(v): k for k, v in _JL_ELTYPE.items()}[tuple(name)]
            return np.frombuffer(doc["data"], dtype=dt.newbyteorder("<")).astype(dt).reshape(size, order="F")
        return [raise_(v) for v in doc["data"]]
    if tag == "symbol":
        return Symbol(doc["name"])
    if tag == "tuple":
        return tuple(raise_(v) for v in doc["data"])
    if tag == "struct" and doc["type"]["name"][-1] == "Dict":
        ks, vs = raise_(doc["data"][0]), raise_(doc["data"][1])
        return dict(zip(ks, vs))
    if tag == "struct" and doc["type"]["name"][-1] == "VersionNumber":
        return "v" + ".".join(str(int(v)) for v in doc["data"][:3])
    return OrderedDict((k, raise_(v)) for k, v in doc.items())


# ---- save_output / load_output (io.jl:70-158, 171-255) ------------------------------------------------------------------
KINETICA_CORE_VERSION = (0, 7, 2)     # Project.toml:4 of the reference snapshot


def _profile_dict(p):
    from . import conditions as C
    if isinstance(p, C.StaticConditionProfile):
        return OrderedDict([("pType", Symbol(type(p).__name__)), ("value", float(p.value))])
    d = OrderedDict()
    for name, val in vars(p).items():                       # fieldnames(pType) .=> getfield (io.jl:92)
        if name == "sol":
            continue
        if callable(val):
            d[name] = None                                  # :f / :grad are not saved (io.jl:100-104)
        elif isinstance(val, np.ndarray):
            d[name] = np.asarray(val, dtype=np.float64)
        elif isinstance(val, (int, float, np.floating, np.integer, bool)) or val is None:
            d[name] = val
    if getattr(p, "sol", None) is not None:
        d["sol"] = OrderedDict([("u", np.asarray(p.sol.u, dtype=np.float64)), ("t", np.asarray(p.sol.t, dtype=np.float64))])
    d["pType"] = Symbol(type(p).__name__)
    return d


def output_tree(out):
    """The `savedict` of save_output (io.jl:108-155) for the host mirror's ODESolveOutput. Fields of the reference's
    containers that the solve path does not carry (xyz geometries, atom-mapped reaction strings, reaction hashes,
    discovery levels) are written as empty collections."""
    sol, pars, cs = out.sol, out.pars, out.conditions
    sol_vcs = None if out.sol_vcs is None else {str(k): np.asarray(v, dtype=np.float64) for k, v in out.sol_vcs.items()}
    sol_k = None if out.sol_k is None else OrderedDict([("u", ("vecvec", np.asarray(out.sol_k.u))), ("t", np.asarray(out.sol_k.t, dtype=np.float64))])
    u0 = pars.u0
    if isinstance(u0, dict):
        u0 = ("dict_sf", dict(u0))
    else:
        u0 = np.asarray(u0, dtype=np.float64)
    lk = pars.low_k_cutoff
    return OrderedDict([
        ("KineticaCoreVersion", ("version", KINETICA_CORE_VERSION)),
        ("sd", OrderedDict([("toInt", ("dict_si", dict(out.sd.toInt))), ("n", int(out.sd.n)), ("xyz", []), ("level_found", ("dict_ii", {}))])),
        ("rd", OrderedDict([("nr", int(out.rd.nr)), ("mapped_rxns", ("vec_string", [])),
                            ("id_reacs", ("vecvec_i", out.rd.id_reacs)), ("id_prods", ("vecvec_i", out.rd.id_prods)),
                            ("stoic_reacs", ("vecvec_i", out.rd.stoic_reacs)), ("stoic_prods", ("vecvec_i", out.rd.stoic_prods)),
                            ("dH", np.asarray(out.rd.dH if out.rd.dH is not None else [], dtype=np.float64)),
                            ("rhash", []), ("level_found", np.ones(out.rd.nr, dtype=np.int64))])),
        ("pars", OrderedDict([("tspan", (float(pars.tspan[0]), float(pars.tspan[1]))), ("u0", u0),
                              ("solver", Symbol("HIPRK45" if pars.explicit else "HIPBDF")), ("jac", bool(pars.jac)), ("sparse", bool(pars.sparse)),
                              ("adaptive_tols", bool(pars.adaptive_tols)), ("update_tols", bool(pars.update_tols)),
                              ("solve_chunks", bool(pars.solve_chunks)), ("solve_chunkstep", float(pars.solve_chunkstep)),
                              ("maxiters", int(pars.maxiters)), ("ban_negatives", bool(pars.ban_negatives)), ("progress", bool(pars.progress)),
                              ("save_interval", None if pars.save_interval is None else float(pars.save_interval)),
                              ("low_k_cutoff", Symbol(lk) if isinstance(lk, str) else float(lk)),
                              ("allow_short_u0", bool(pars.allow_short_u0))])),
        ("sol", OrderedDict([("u", ("vecvec", sol.u)), ("t", np.asarray(sol.t, dtype=np.float64)), ("vcs", sol_vcs), ("k", sol_k)])),
        ("conditions", OrderedDict([("symbols", ("vec_symbol", [Symbol(s) for s in cs.symbols])), ("profiles", [_profile_dict(p) for p in cs.profiles]),
                                    ("discrete_updates", bool(cs.discrete_updates)),
                                    ("ts_update", None if cs.ts_update is None else float(cs.ts_update))])),
    ])


def _lower_tree(x):
    if isinstance(x, tuple) and len(x) == 2 and isinstance(x[0], str):
        kind, v = x
        if kind == "vecvec":
            return lower_vector_of_vectors(np.asarray(v, dtype=np.float64), np.float64)
        if kind == "vecvec_i":
            return lower_vector_of_vectors([np.asarray(r, dtype=np.int64) for r in v], np.int64)
        if kind == "vec_symbol":                      # Vector{Symbol}
            return lower_typed_vector(v, ["Core", "Symbol"])
        if kind == "vec_string":                      # Vector{String}
            return lower_typed_vector(v, ["Core", "String"])
        if kind == "dict_si":
            return lower_dict(v, lower_datatype(["Core", "String"]), lower_datatype(["Core", "Int64"]))
        if kind == "dict_sf":
            return lower_dict(v, lower_datatype(["Core", "String"]), lower_datatype(["Core", "Float64"]))
        if kind == "dict_ii":
            return lower_dict(v, lower_datatype(["Core", "Int64"]), lower_datatype(["Core", "Int64"]))
        if kind == "version":
            return OrderedDict([("tag", "struct"), ("type", lower_datatype(["Base", "VersionNumber"])),
                                ("data", [int(v[0]), int(v[1]), int(v[2]), lower_tuple(()), lower_tuple(())])])
    if isinstance(x, dict):
        return OrderedDict((k, _lower_tree(v)) for k, v in x.items())
    if isinstance(x, list):
        return [_lower_tree(v) for v in x]
    return lower(x)


def save_output(out, saveto):
    """save_output(out::ODESolveOutput, saveto::String) (io.jl:70-158)."""
    with open(saveto, "wb") as f:
        f.write(dumps(_lower_tree(output_tree(out))))


def load_output(outfile):
    """load_output(outfile) (io.jl:171-255): rebuilds an ODESolveOutput of the host mirror from the dictionary tree.
    As in the reference some data is lost: profile functions come back as stubs that raise, the solution is a plain
    (t, u) array pair with linear interpolation (DiffEqArray)."""
    from . import conditions as C
    from . import solving as S
    tree = raise_(loads(open(outfile, "rb").read()))
    sd_t, rd_t, p_t, s_t, c_t = tree["sd"], tree["rd"], tree["pars"], tree["sol"], tree["conditions"]
    toInt = {str(k): int(v) for k, v in sd_t["toInt"].items()}
    # the profile type named in the file selects a constructor from THIS list only (the reference evaluates the name,
    # io.jl:238: a file is data, not code)
    profile_types = {c.__name__: c for c in (C.StaticConditionProfile, C.NullDirectProfile, C.LinearDirectProfile,
                                             C.NullGradientProfile, C.LinearGradientProfile, C.DoubleRampGradientProfile)}
    sd = S.SpeciesData(toInt, {v: k for k, v in toInt.items()}, int(sd_t["n"]))
    as_rows = lambda rows: [[int(x) for x in r] for r in rows]
    dH = np.asarray(rd_t["dH"], dtype=float)
    rd = S.RxData(int(rd_t["nr"]), as_rows(rd_t["id_reacs"]), as_rows(rd_t["id_prods"]), as_rows(rd_t["stoic_reacs"]),
                  as_rows(rd_t["stoic_prods"]), list(dH) if dH.size else None)
    u0 = p_t["u0"]
    u0 = {str(k): float(v) for k, v in u0.items()} if isinstance(u0, dict) else np.asarray(u0, dtype=float)
    lk = p_t["low_k_cutoff"]
    pars = S.ODESimulationParams(tspan=tuple(float(x) for x in p_t["tspan"]), u0=u0,
                                 solver="RK45" if str(p_t["solver"]) == "HIPRK45" else None, jac=p_t["jac"], sparse=p_t["sparse"],
                                 adaptive_tols=p_t["adaptive_tols"], update_tols=p_t["update_tols"], solve_chunks=p_t["solve_chunks"],
                                 solve_chunkstep=p_t["solve_chunkstep"], maxiters=p_t["maxiters"], ban_negatives=p_t["ban_negatives"],
                                 progress=p_t["progress"], save_interval=p_t["save_interval"],
                                 low_k_cutoff=str(lk) if isinstance(lk, str) else float(lk), allow_short_u0=p_t["allow_short_u0"])
    t = np.asarray(s_t["t"], dtype=float)
    u = np.array([np.asarray(r, dtype=float) for r in s_t["u"]]).reshape(len(t), -1)
    sol_k = None if s_t["k"] is None else S.DiscreteRates(np.asarray(s_t["k"]["t"], dtype=float),
                                                          np.array([np.asarray(r, dtype=float) for r in s_t["k"]["u"]]))
    sol_vcs = None if s_t["vcs"] is None else {k: np.asarray(v, dtype=float) for k, v in s_t["vcs"].items()}
    profiles = {}
    for sym, pd in zip(c_t["symbols"], c_t["profiles"]):
        if str(pd["pType"]) not in profile_types:
            raise ValueError(f"unknown condition profile type {pd['pType']!r} in {outfile}")
        cls = profile_types[str(pd["pType"])]
        if issubclass(cls, C.StaticConditionProfile):
            profiles[str(sym)] = cls(pd["value"])
            continue
        p = cls.__new__(cls)                                  # eval(Meta.parse(String(pType)))(values(profile_dict)...)
        for k, v in pd.items():
            if k in ("pType", "sol"):
                continue
            setattr(p, k, _loaded_profile_null_func if v is None and k in ("f", "grad") else v)
        p.sol = None if "sol" not in pd else C.ProfileSolution(np.asarray(pd["sol"]["t"], dtype=float), np.asarray(pd["sol"]["u"], dtype=float))
        profiles[str(sym)] = p
    cs = C.ConditionSet(profiles, ts_update=c_t["ts_update"])
    sol = S.ODESolution(t, u, "Success", k=sol_k)
    return S.ODESolveOutput(sd, rd, sol, sol_k, sol_vcs, pars, cs)


def _loaded_profile_null_func(*a, **k):
    raise RuntimeError("Condition profile function saving is not supported. Condition function must be manually rebuilt.")
