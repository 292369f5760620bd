"""the soft-register (CSR) codec and the ddt_cli host program that consumes the reference's
register block + line-stream files."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
import ddt


def _encode(p, n, devices=1):
    csr = (C.c_uint64 * 12)()
    rc = ddt.lib().ddt_csr_encode(C.byref(p), n, devices, C.byref(csr))
    return rc, csr


def test_csr_fields_sit_where_the_rtl_reads_them():
    # BASELINE config 3 on one device: field positions of rtl/DTEngine/EngineCSR.sv:190-248
    p = ddt.make_params(1000, 8, 32)
    rc, csr = _encode(p, 100_000_000)
    assert rc == 0
    assert csr[0] & 1                                                       # 200: start
    assert csr[2] & 0xFFFFFFFF == 1000 * (128 + 32) and csr[2] >> 32 == 1000 * 128   # 202: total / weights lines
    assert (csr[4] >> 16) & 0xFFFF == 128 and (csr[4] >> 32) & 0xFFFF == 32 and csr[4] >> 48 == 8   # 204
    assert csr[4] & 0xFF == 0x01 and (csr[4] >> 8) & 0xFF == 0xFF           # C = 8: one replica, all clusters per tuple
    assert csr[5] & 0xFFFFFFFF == 0x7FC00000 and (csr[5] >> 32) & 0xF == 8  # 205: missing, levels
    assert (csr[5] >> 36) & 0xFF == 16 and (csr[5] >> 44) & 0xF == 8        #      16 trees per PU, 8 clusters per tuple
    assert csr[7] & 0xFFFFFFFF == 25_000_000                                # 207: N / 4 result lines
    assert (csr[3] >> 32) & 0xFF == 1


@pytest.mark.parametrize("T,D,F,n,dev", [(8, 4, 16, 1000, 1), (100, 6, 28, 10_000_001, 1), (1000, 8, 32, 4096, 8),
                                         (3, 16, 2048, 7, 1), (512, 12, 64, 12, 1)])
def test_csr_roundtrip(T, D, F, n, dev):
    p = ddt.make_params(T, D, F, missing_bits=0x12345678, clusters=ddt.default_clusters((T + dev - 1) // dev))
    rc, csr = _encode(p, n, dev)
    if D == 12 and T == 512:
        assert rc == 0
    assert rc == 0
    q, n2, d2 = ddt.Params(), C.c_uint64(), C.c_uint32()
    assert ddt.lib().ddt_csr_decode(C.byref(csr), C.byref(q), C.byref(n2), C.byref(d2)) == 0
    assert (q.num_trees, q.num_levels, q.missing_bits) == (T, D, 0x12345678)
    assert q.num_features == ddt.tuple_words(F) and q.clusters_per_tuple == p.clusters_per_tuple
    assert (q.weights_lines_per_tree, q.findex_lines_per_tree) == (p.weights_lines_per_tree, p.findex_lines_per_tree)
    assert n2.value == (n + 3) // 4 * 4 and d2.value == dev


def _rtl_device_list(csr):
    """devices_list[] exactly as rtl/DTEngine/EngineCSR.sv:250-296 slices registers 208-210 (5 bits at a byte stride;
    register 210 holds ids 16..19 only)."""
    out = []
    for i in range(20):
        reg = csr[8 + i // 8]
        lo = 8 * (i % 8)
        out.append((reg >> lo) & 0x1F)
    return out


@pytest.mark.parametrize("devices", [1, 2, 8, 9, 20])
def test_csr_device_list_sits_where_the_rtl_slices_it(devices):
    p = ddt.make_params(40, 4, 16)
    rc, csr = _encode(p, 4096, devices)
    assert rc == 0
    lst = _rtl_device_list(csr)
    assert lst[:devices] == list(range(devices)) and not any(lst[devices:])
    assert csr[10] >> 32 == 0  # CSR 210: ids 16..19 only
    assert (csr[3] >> 32) & 0xFF == devices
    # the library's own decoder slices the same way
    ids = (C.c_uint8 * 20)()
    q, mode, flags = ddt.Params(), C.c_uint32(), C.c_uint32()
    assert ddt.lib().ddt_csr_decode_ex(C.byref(csr), C.byref(q), None, None, C.byref(mode), C.byref(flags), C.byref(ids)) == 0
    assert list(ids) == lst and mode.value == 0


def test_csr_mode_flags_per_device_for_both_sharding_modes():
    """CSR 201 (EngineCSR.sv:194-205): [0] data_distributed [1] host_node [2] broadcast_data [3] broadcast_trees
    [4] aggreg_enabled [5] multiple_nodes [6] pcie_receiver_enabled [7] last_node; the two modes of DTInference.sv:28-37."""
    L = ddt.lib()
    p = ddt.make_params(64, 6, 28)
    n = 1000

    def enc(mode, dev, G):
        csr = (C.c_uint64 * 12)()
        assert L.ddt_csr_encode_ex(C.byref(p), n, G, mode, dev, C.byref(csr)) == 0
        return csr

    HOST, BDATA, BTREES, AGG, MULTI, PCIE, LAST = 1 << 1, 1 << 2, 1 << 3, 1 << 4, 1 << 5, 1 << 6, 1 << 7
    # tree-sharded, 4 devices: tuples broadcast + partial results aggregated on every device
    for d in range(4):
        f = enc(0, d, 4)[1] & 0xFF
        assert f & (BDATA | AGG | MULTI) == BDATA | AGG | MULTI and not f & BTREES
        assert bool(f & HOST) == (d == 0) and bool(f & PCIE) == (d == 0) and bool(f & LAST) == (d == 3)
    # row-sharded: trees broadcast, tuples dealt in batches of 4 (= 4 * tuple lines), no aggregation
    for d in range(4):
        csr = enc(1, d, 4)
        f = csr[1] & 0xFF
        assert f & (BTREES | MULTI) == BTREES | MULTI and not f & (BDATA | AGG)
        assert bool(f & HOST) == (d == 0) and bool(f & LAST) == (d == 3)
        assert csr[1] >> 32 == 4 * 7  # 28 features = 7 lines per tuple, 4 tuples per batch (DTInference.sv:35-36)
        assert (csr[5] >> 36) & 0xFF == 8  # every device holds all 64 trees: 8 groups / 1 cluster
        mode = C.c_uint32()
        q = ddt.Params()
        assert L.ddt_csr_decode_ex(C.byref(csr), C.byref(q), None, None, C.byref(mode), None, None) == 0 and mode.value == 1
    # result lines each device announces (CSR 207): the host sees all, the others their round-robin share
    lines = [enc(1, d, 4)[7] for d in range(4)]
    assert lines[0] == 250 and sum(lines[1:]) + (250 // 4 + (1 if 0 < 250 % 4 else 0)) == 250
    # one device: host and last at once, nothing broadcast
    f = enc(0, 0, 1)[1] & 0xFF
    assert f == HOST | PCIE | LAST
    # next-hop addresses close the ring (CSR 206 [7:0] broadcast, [15:8] results)
    assert [enc(0, d, 3)[6] & 0xFF for d in range(3)] == [1, 2, 0]
    assert L.ddt_csr_encode_ex(C.byref(p), n, 4, 2, 0, C.byref((C.c_uint64 * 12)())) == -1   # unknown mode
    assert L.ddt_csr_encode_ex(C.byref(p), n, 4, 0, 4, C.byref((C.c_uint64 * 12)())) == -1   # device index out of range


def test_csr_rejects_inconsistent_blocks():
    p = ddt.make_params(8, 4, 16)
    _, csr = _encode(p, 100)
    q = ddt.Params()
    bad = (C.c_uint64 * 12)(*csr)
    bad[2] += 1  # total lines no longer T * (wl + fl)
    assert ddt.lib().ddt_csr_decode(C.byref(bad), C.byref(q), None, None) == -1
    bad = (C.c_uint64 * 12)(*csr)
    bad[5] = (bad[5] & ~(0xF << 44)) | (3 << 44)  # 3 clusters per tuple
    assert ddt.lib().ddt_csr_decode(C.byref(bad), C.byref(q), None, None) == -1
    # 250 trees x 512 lines per device do not fit the RTL's 16-bit per-device line counter (EngineCSR.sv:213)
    assert ddt.lib().ddt_csr_encode(C.byref(ddt.make_params(1000, 10, 32)), 8, 4, C.byref(csr)) == -5


def test_cli_gen_writes_wire_format_files(tmp_path):
    assert os.path.exists(ddt.CLI_PATH), "ddt_cli not built"
    pre = str(tmp_path / "m")
    subprocess.check_call([ddt.CLI_PATH, "gen", "--trees", "100", "--levels", "6", "--features", "28", "--rows", "515",
                           "--dist", "1", "--prefix", pre])
    m = O.gen_model(100, 6, 28, dist=1)
    assert np.array_equal(np.fromfile(pre + ".weights", np.uint32), m.wlines)
    assert np.array_equal(np.fromfile(pre + ".findex", np.uint16), m.flines)
    assert np.array_equal(np.fromfile(pre + ".tuples", np.uint32).reshape(515, 28), O.gen_tuples(0, 515, 28, dist=1))
    regs = dict((int(a), int(v, 16)) for a, v in (l.split() for l in open(pre + ".csr") if l[0] != "#"))
    assert regs[202] & 0xFFFFFFFF == 100 * (32 + 8) and regs[207] == (515 + 3) // 4 and list(regs)[-1] == 200


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,F,n,dist", [(8, 4, 16, 1000, 0), (1000, 8, 32, 3001, 0), (100, 6, 28, 2222, 1)])
def test_cli_score_matches_oracle(tmp_path, T, D, F, n, dist):
    pre = str(tmp_path / "m")
    subprocess.check_call([ddt.CLI_PATH, "gen", "--trees", str(T), "--levels", str(D), "--features", str(F),
                           "--rows", str(n), "--dist", str(dist), "--prefix", pre])
    out = subprocess.check_output([ddt.CLI_PATH, "score", "--csr", pre + ".csr", "--weights", pre + ".weights",
                                   "--findex", pre + ".findex", "--tuples", pre + ".tuples", "--out", pre + ".results"]).decode()
    assert f"scored {n} tuples" in out
    res = np.fromfile(pre + ".results", np.float32)
    assert res.size == (n + 3) // 4 * 4 and not res[n:].any()  # whole 128-bit result lines (ResultsCombiner.sv:136-160)
    m = O.gen_model(T, D, F, dist=dist)
    want = O.score(m, O.gen_tuples(0, n, F, dist=dist))
    assert np.array_equal(res[:n].view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_cli_scores_a_sparse_forest(tmp_path):
    """gen-sparse / score-sparse: the explicit-children stream as files, scored by the C++ host alone."""
    pre = str(tmp_path / "rf")
    T, D, F, n = 24, 13, 20, 2051
    subprocess.check_call([ddt.CLI_PATH, "gen-sparse", "--trees", str(T), "--max-depth", str(D), "--features", str(F), "--rows", str(n),
                           "--full-levels", "4", "--permille", "650", "--dist", "1", "--prefix", pre])
    s = O.gen_sparse_model(T, D, F, 4, 650, 1)
    assert np.array_equal(np.fromfile(pre + ".nodes", np.uint32).reshape(-1, 4), s.node_lines)
    assert np.array_equal(np.fromfile(pre + ".first", np.uint64), s.first)
    out = subprocess.check_output([ddt.CLI_PATH, "score-sparse", "--nodes", pre + ".nodes", "--first", pre + ".first", "--tuples", pre + ".tuples",
                                   "--features", str(F), "--max-depth", str(D), "--out", pre + ".results"]).decode()
    assert f"scored {n} tuples with sparse trees [0, {T})" in out and "sparse_" in out
    res = np.fromfile(pre + ".results", np.float32)
    want = O.score_sparse(s, O.gen_tuples(0, n, F, dist=1))
    assert res.size == (n + 3) // 4 * 4 and np.array_equal(res[:n].view(np.uint32), want.view(np.uint32))


def test_cli_rejects_a_bad_rank_before_touching_a_device(tmp_path):
    """--ranks / --rank (one process per GPU, ddt_comm_*) are checked before ddt_create: no GPU needed for the refusal."""
    pre = str(tmp_path / "m")
    subprocess.check_call([ddt.CLI_PATH, "gen", "--trees", "16", "--levels", "4", "--features", "16", "--rows", "8", "--prefix", pre])
    base = [ddt.CLI_PATH, "score", "--csr", pre + ".csr", "--weights", pre + ".weights", "--findex", pre + ".findex", "--tuples", pre + ".tuples"]
    for extra in (["--ranks", "2", "--rank", "2", "--id-file", pre + ".id"], ["--ranks", "0", "--rank", "0", "--id-file", pre + ".id"],
                  ["--ranks", "2", "--rank", "0", "--id-file", pre + ".id", "--combine", "tree"]):
        r = subprocess.run(base + extra, capture_output=True)
        assert r.returncode == 1 and b"--ranks / --rank / --combine" in r.stderr
    r = subprocess.run(base + ["--ranks", "2"], capture_output=True)          # --rank is required
    assert r.returncode == 2 and b"missing --rank" in r.stderr


def test_csr_blocks_round_trip_and_corrupted_blocks_are_refused_or_decoded():
    """Encode -> decode recovers what the wire carries (tuple lines x 4 features, result lines x 4 tuples) for random run parameters;
    a block with one field flipped or replaced decodes to something or is refused with DDT_EINVAL, never anything else."""
    L = ddt.lib()
    A = C.c_uint64 * 12
    rng = np.random.default_rng(7)
    seen = set()
    for _ in range(4000):
        T, D, F = int(rng.integers(1, 3000)), int(rng.integers(1, 12)), int(rng.integers(1, 2049))
        p, csr = ddt.make_params(T, D, F), A()
        n, G = int(rng.integers(0, 1 << 33)), int(rng.integers(1, 21))
        rc = L.ddt_csr_encode(C.byref(p), n, G, C.byref(csr))
        assert rc in (0, -5), rc                                   # -5: the count does not fit CSR207's 32 bits of result lines
        if rc:
            continue
        mode, k = int(rng.integers(0, 3)), int(rng.integers(0, 12))
        if mode == 1:
            csr[k] ^= 1 << int(rng.integers(0, 64))
        elif mode == 2:
            csr[k] = int(rng.integers(0, 1 << 63))
        q, nt, nd = ddt.Params(), C.c_uint64(), C.c_uint32()
        rc = L.ddt_csr_decode(C.byref(csr), C.byref(q), C.byref(nt), C.byref(nd))
        assert rc in (0, -1), rc
        seen.add((mode, rc))
        if mode == 0:
            assert rc == 0 and (q.num_trees, q.num_levels, q.num_features, nt.value, nd.value) == (T, D, (F + 3) // 4 * 4, (n + 3) // 4 * 4, G)
    assert {(0, 0), (1, 0), (1, -1), (2, -1)} <= seen
