"""CPU replay of the tiled sweep's arithmetic from the library-order tables (test infrastructure).

Follows kinetica_jl_amd/csrc/tiled_kernels.hip step by step - hubs and one window at a time in an emulated LDS image,
dummy entries, split accumulators, the per-64-record flag bits that let a wavefront skip an unused field - so that the
tables built by tiled.cpp can be checked against the oracle without a GPU.
"""
import numpy as np


def replay(L, u_lib, k_lib):
    """du_lib for one state. L = capi.lib_layout_host(net); u_lib[N], k_lib[k_len] in library order."""
    h, T, E, wbase, BS, n_copy = L["h"], L["T"], L["E"], L["wbase"], L["BS"], L["n_copy"]
    N = len(u_lib)
    du_lib = np.full(N, np.nan)
    u_s = np.full(E, np.nan)
    du_s = np.zeros(E)
    u_s[h:h + 64] = 1.0
    u_s[:h] = u_lib[:h]
    for c in range(n_copy):
        u_s[h + 64 + c] = u_lib[L["copy_src"][c]]
    rec = L["rec"]
    for s in range(T):
        off, cnt = int(L["win_off"][s]), int(L["win_cnt"][s])
        koff, n2 = int(L["seg_k"][s, 0]), int(L["seg_k"][s, 1])
        seg_first = int(L["rowtab"][int(L["seg_q"][s]), 0])
        assert koff % 2 == 0 and (n2 % 64 == 0 or seg_first < 0 or n2 == int(L["rowtab"][int(L["seg_q"][s]):int(L["seg_q"][s + 1]), 1].sum()))
        u_s[wbase:wbase + cnt] = u_lib[off:off + cnt]
        for q in range(int(L["seg_q"][s]), int(L["seg_q"][s + 1])):
            first, n = int(L["rowtab"][q, 0]), int(L["rowtab"][q, 1])
            if first < 0:
                continue
            assert 0 < n <= BS
            w = rec[first:first + n]
            fl = (w >> np.uint64(56)).astype(np.int64)
            # the kernel takes a wavefront's flags from its first lane: they must be equal across each group of 64
            for g in range(0, n, 64):
                assert np.all(fl[g:g + 64] == fl[g]), "flags differ inside a wavefront's records"
            assert not np.any(fl & 4)
            l = [((w >> np.uint64(14 * j)) & np.uint64(0x3fff)).astype(np.int64) for j in range(4)]
            assert max(x.max() for x in l) < E
            has1, has3 = (fl & 1) != 0, (fl & 2) != 0
            # a record that uses a field lies in a group whose flag for it is set
            uf = u_s[l[0]] * np.where(has1, u_s[l[1]], 1.0)
            ur = u_s[l[2]] * np.where(has3, u_s[l[3]], 1.0)
            # rate constants: two slots per record below n2 (position in the segment), one from there on (tiled.hpp)
            i = first - seg_first + np.arange(n)
            two = i < n2
            kf = k_lib[koff + np.where(two, 2 * i, n2 + i)]
            kr = np.where(two, k_lib[np.minimum(koff + 2 * i + 1, len(k_lib) - 1)], 0.0)
            net = kf * uf - kr * ur
            assert np.all(np.isfinite(net))
            np.add.at(du_s, l[0], -net)
            np.add.at(du_s, l[1][has1], -net[has1])
            np.add.at(du_s, l[2], net)
            np.add.at(du_s, l[3][has3], net[has3])
        if s == T - 1:
            for c in range(n_copy):
                du_s[L["copy_src"][c]] += du_s[h + 64 + c]
                du_s[h + 64 + c] = 0.0
        du_lib[off:off + cnt] = du_s[wbase:wbase + cnt]
        du_s[wbase:wbase + cnt] = 0.0
        u_s[wbase:] = np.nan
    du_lib[:h] = du_s[:h]
    return du_lib


def k_to_lib(L, k):
    out = np.zeros(L["k_len"])
    out[L["slot_of_reaction"]] = k
    return out
