"""The per-device split of the model and tuple streams, pinned to the reference's own RTL.

tests/golden/receiver_rtl_vectors.npz was produced by tests/golden/make_receiver_golden.py, which EXECUTES
rtl/DTEngine/PCIeReceiver.sv:136-150,156-180,186-316 (the host node's stream router: line stamp, local / SL3 routing, the
FSM with its line counters and running device index) on the registers that EngineCSR.sv:146-308 -- executed too -- derives
from the CSR blocks of the PRODUCT's codec (ddt_csr_encode_ex).  Per stream line it holds {FSM state, prog_mode,
data_valid, device index, stays-local}.

Held to it: the product's tree -> device map (ddt_shard_range: the shards ddt_load_model_shard / ddt_comm / ddt_group
load; Python mirror ddt.shard_bounds), the oracle's multi-device model (orc_score with n_devices), the stamp the
PU programming golden assumes (weights lines first with prog_mode 1, then feature-index lines with 0, tuples with
data_valid 1), and the row mode's dealing of tuples.  One more defect of the published RTL is asserted as recorded."""
import ctypes as C
import os

import numpy as np
import pytest

import ddt
from ddt import shard_bounds
from oracle import oracle as O

VEC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "receiver_rtl_vectors.npz")
STATE, PROG, VALID, DEV, LOCAL = range(5)
RECEIVE_TREES, RECEIVE_DATA = 1, 3


@pytest.fixture(scope="module")
def vec():
    return np.load(VEC)


def _cases(v):
    for row in v["cases"]:
        T, D, F, G, mode, pad, wl, fl, tl, n = (int(x) for x in row)
        rec = v[f"rec_{T}_{D}_{F}_{G}_{mode}_{pad}"]
        yield dict(T=T, D=D, F=F, G=G, mode=mode, wl=wl, fl=fl, tl=tl, n=n, w=rec[:T * wl], f=rec[T * wl:T * (wl + fl)],
                   x=rec[T * (wl + fl):], csr=v[f"csr_{T}_{D}_{F}_{G}_{mode}_{pad}"], last=v[f"last_{T}_{D}_{F}_{G}_{mode}_{pad}"])


def _product_owner(T, G):
    owner = np.full(T, -1)
    for g in range(G):
        b, e = C.c_uint32(), C.c_uint32()
        assert ddt.lib().ddt_shard_range(T, g, G, C.byref(b), C.byref(e)) == 0
        assert (b.value, e.value) == shard_bounds(T, G)[g]
        owner[b.value:e.value] = g
    assert (owner >= 0).all()
    return owner


def test_the_stream_is_stamped_weights_first_then_feature_indexes_then_tuples(vec):
    for c in _cases(vec):
        assert (c["w"][:, STATE] == RECEIVE_TREES).all() and (c["w"][:, PROG] == 1).all() and not c["w"][:, VALID].any()
        assert (c["f"][:, STATE] == RECEIVE_TREES).all() and not c["f"][:, PROG].any() and not c["f"][:, VALID].any()
        assert (c["x"][:, STATE] == RECEIVE_DATA).all() and (c["x"][:, VALID] == 1).all()
        # a line stays on the host device exactly when the running device index is 0 (PCIeReceiver.sv:156-180)
        for part in (c["w"], c["f"], c["x"]):
            assert ((part[:, DEV] == 0) == (part[:, LOCAL] == 1)).all()


def test_tree_shards_of_the_product_are_the_rtls(vec):
    seen = 0
    for c in _cases(vec):
        if c["mode"] != 0:
            continue
        T, G, wl, fl = c["T"], c["G"], c["wl"], c["fl"]
        owner = _product_owner(T, G)
        w_dev = c["w"][:, DEV].reshape(T, wl)
        assert (w_dev == w_dev[:, :1]).all(), "a tree's weights lines all go to one device"
        assert (w_dev[:, 0] == owner).all(), (T, G)              # contiguous ceil(T/G) shards in device-list order
        f_dev = c["f"][:, DEV].reshape(T, fl)
        assert (f_dev == f_dev[:, :1]).all()
        if T % G == 0:
            assert (f_dev[:, 0] == owner).all(), (T, G)          # feature indexes follow their trees
        else:
            # defect (6) of the published RTL: the device index is not reset between the two streams, so the feature-index
            # stream starts on the device the weights stream stopped at -- the product keeps a tree's two halves together
            start = int(w_dev[-1, 0]) if int((w_dev[:, 0] == w_dev[-1, 0]).sum()) < -(-T // G) else (int(w_dev[-1, 0]) + 1) % G
            assert int(f_dev[0, 0]) == start != 0 and not (f_dev[:, 0] == owner).all(), (T, G)
        assert not c["x"][:, DEV].any()                          # tuples: all to the host device, which re-broadcasts them
        seen += 1
    assert seen >= 8


def test_row_mode_keeps_the_model_whole_and_deals_tuples_in_batches_of_four(vec):
    seen = 0
    for c in _cases(vec):
        if c["mode"] != 1:
            continue
        assert not c["w"][:, DEV].any() and not c["f"][:, DEV].any()        # broadcast_trees: every line enters the host device
        batch = int(c["csr"][1]) >> 32                                      # CSR201[63:32]: lines per batch
        assert batch == 4 * c["tl"]
        t_dev = c["x"][:, DEV].reshape(c["n"], c["tl"])
        assert (t_dev == t_dev[:, :1]).all()
        assert (t_dev[:, 0] == (np.arange(c["n"]) // 4) % c["G"]).all()     # PCIeReceiver.sv:289-312
        seen += 1
    assert seen >= 3


@pytest.mark.parametrize("T,G,C_", [(16, 2, 1), (24, 8, 1), (40, 4, 2), (37, 8, 4), (9, 4, 1)])
def test_oracle_multi_device_model_uses_the_same_split(T, G, C_):
    """orc_score(n_devices = G) == per-device partial sums over the RTL's shards, chain-added host -> dev1 -> ..."""
    D, F = 3, 8
    m = O.gen_model(T, D, F, 1, clusters=C_)
    x = O.gen_tuples(0, 64, F, 1)
    want = O.score(m, x, n_devices=G)
    owner = _product_owner(T, G)
    acc = None
    for g in range(G):
        idx = np.nonzero(owner == g)[0]
        part = (O.score_shard(m, x, int(idx[0]), int(idx[-1]) + 1) if idx.size else np.zeros(64, np.float32))
        acc = part if acc is None else O.fpadd_bits_batch(acc.view(np.uint32), part.view(np.uint32)).view(np.float32)
    assert np.array_equal(acc.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("G", [2, 3, 5, 8])
def test_result_chain_adds_in_device_list_order_and_the_host_forwards(vec, G):
    """ResultsCombiner.sv:355-393,422-453 executed device by device (+ the pinned hop adders): the host puts its local line on the
    link unchanged, device d sends local + upstream, the host hands what comes back to PCIe as it is -- p0 + p1 + ... + p(G-1) in
    device-list order, which is what orc_score(n_devices), DDT_COMBINE_CHAIN and ddt_chain_sum_device compute."""
    parts, final, clean = vec[f"chain_parts_{G}"], vec[f"chain_final_{G}"], vec[f"chain_clean_{G}"]
    acc = parts[0].reshape(-1)
    for d in range(1, G):
        acc = O.fpadd_bits_batch(acc, parts[d].reshape(-1))
    acc = acc.reshape(final.shape)
    assert clean.mean() > 0.9                                    # words with an exact cancellation on some hop are the RTL's defect
    assert np.array_equal(acc[clean], final[clean])


def test_last_is_stamped_on_the_final_line_of_every_tree_and_tuple(vec):
    """InputDistributor.sv:247-288 executed on the receiver's stamps and the executed registers: the local core sees `last` on line
    lines-per-tree - 1 of every tree's weights, of every tree's feature indexes and on the final line of every tuple -- the
    boundaries the PU programming vectors (tests/test_oracle_program.py) are driven with.  Here the *_minus_one registers are used
    as a compare value, correctly: defect 1 is confined to the control word's stride fields."""
    for c in _cases(vec):
        T, wl, fl, tl, n = c["T"], c["wl"], c["fl"], c["tl"], c["n"]
        w, f, x = c["last"][:T * wl].reshape(T, wl), c["last"][T * wl:T * (wl + fl)].reshape(T, fl), c["last"][T * (wl + fl):].reshape(n, tl)
        for part in (w, f, x):
            assert (part[:, -1] == 1).all() and not part[:, :-1].any()


def test_ring_rebroadcast_reaches_every_device_once(vec):
    """InputDistributor.sv:199-232 executed: in the tree-sharded mode (broadcast_data) a tuple line goes to the local core AND on
    to the next device unless this is the last node; tree lines stay local (the receiver already routed them).  With the mode
    flags the codec writes per device (host, ..., last) every device of the list therefore sees every tuple line exactly once --
    what "tuples replicated on every rank" (and tuples_to_device's hand-over) provide.  Row mode: the trees travel, the data stays."""
    route = {tuple(int(v) for v in r[:4]): (int(r[4]), int(r[5])) for r in vec["route"]}     # (data line, bcast_data, bcast_trees, last) -> (core, next)
    for last_node in (0, 1):
        assert route[(1, 1, 0, last_node)] == (1, 1 - last_node)         # tree-sharded mode, tuple line
        assert route[(0, 1, 0, last_node)] == (1, 0)                     #                    tree line
        assert route[(0, 0, 1, last_node)] == (1, 1 - last_node)         # row mode, tree line
        assert route[(1, 0, 1, last_node)] == (1, 0)                     #           tuple line: scored where the receiver sent it
    for G in (2, 5, 8):                                                  # walk the ring with the per-device flags of ddt_csr_encode_ex
        p = ddt.make_params(64, 4, 16)
        seen = []
        for d in range(G):
            buf = (C.c_uint64 * 12)()
            assert ddt.lib().ddt_csr_encode_ex(C.byref(p), 1000, G, 0, d, C.byref(buf)) == 0
            flags = int(buf[1]) & 0xFF
            core, nxt = route[(1, (flags >> 2) & 1, (flags >> 3) & 1, (flags >> 7) & 1)]
            seen.append(core)
            if not nxt:
                assert d == G - 1, "the ring must carry the line to the last device of the list"
                break
        assert seen == [1] * G
