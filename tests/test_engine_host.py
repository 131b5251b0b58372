"""CPU: the native index build / parse of an engine step (csrc/engine_host.cpp: aic_step_build, aic_step_parse) against
the numpy expressions they replace (HotPathEngine._begin_numpy / the numpy branch of finish()), on seeded random engine
states: every section of both staging buffers, the totals, the committed token rows."""
import ctypes

import numpy as np
import pytest

from arcticinference_amd import _native as N

MAX_SPEC_LEN = 32


def _numpy_build(live, num_tokens, n_draft_all, draft_ids, draft_row, lstm_k, bt_host, bs, G):
    n_draft = n_draft_all[live]
    B = len(live)
    q_len = n_draft + 1
    T = int(q_len.sum())
    qsl = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(q_len, out=qsl[1:])
    ntok = num_tokens[live]
    ctx = ntok + n_draft
    rep = np.repeat(np.arange(B), q_len)
    pos = (ntok.astype(np.int64) - 1)[rep] + (np.arange(T) - qsl[:-1][rep])
    slot_map = bt_host[live[rep], pos // bs].astype(np.int64) * bs + pos % bs
    is_short = q_len * G <= 32
    order = np.concatenate([np.nonzero(is_short)[0], np.nonzero(~is_short)[0]]).astype(np.int32)
    draft_flat = draft_ids[live][np.arange(MAX_SPEC_LEN)[None, :] < n_draft[:, None]]
    cu_draft = np.cumsum(n_draft).astype(np.int32)
    fill_pos = fill_src = np.zeros(0, np.int64)
    if draft_row is not None:
        pend_rows = draft_row[live]
        sel = np.nonzero(pend_rows >= 0)[0]
        if len(sel):
            k_sel = n_draft[sel].astype(np.int64)
            rp = np.repeat(np.arange(len(sel)), k_sel)
            within = np.arange(int(k_sel.sum())) - np.repeat(np.cumsum(k_sel) - k_sel, k_sel)
            fill_pos = (cu_draft[sel] - k_sel)[rp] + within
            fill_src = pend_rows[sel][rp] * lstm_k + within
    is_bonus = np.zeros(T, dtype=bool)
    is_bonus[qsl[1:] - 1] = True
    return dict(ctx=ctx.astype(np.int32), qsl=qsl, live=live, slots=slot_map, order=order, n_short=int(is_short.sum()),
                draft_flat=draft_flat.astype(np.int32), cu_draft=cu_draft, target_rows=np.nonzero(~is_bonus)[0],
                bonus_rows=(qsl[1:] - 1).astype(np.int64), fill_pos=fill_pos, fill_src=fill_src, T=T,
                max_q=int(q_len.max()), max_ctx=int(ctx.max()), ctx_sum=int(ctx.sum()))


@pytest.mark.parametrize("seed", range(6))
def test_step_build_equals_the_numpy_expressions(seed):
    rng = np.random.default_rng(seed)
    max_seqs, bs, max_len = 64, 16, 600
    bps = (max_len + bs - 1) // bs
    G = [4, 4, 1, 8, 2, 4][seed]
    lstm_k = 3
    num_tokens = rng.integers(1, max_len - 40, max_seqs).astype(np.int32)
    n_draft_all = rng.choice([0, 0, 0, 3, 3, 5, 7, 8, 15, 32], max_seqs).astype(np.int32)
    draft_ids = rng.integers(0, 100000, (max_seqs, MAX_SPEC_LEN)).astype(np.int32)
    draft_row = np.full(max_seqs, -1, np.int64)
    pend = rng.random(max_seqs) < 0.3
    draft_row[pend & (n_draft_all == 3)] = rng.integers(0, 64, int((pend & (n_draft_all == 3)).sum()))
    bt_host = rng.permutation(max_seqs * bps).reshape(max_seqs, bps).astype(np.int32)
    live = np.sort(rng.choice(max_seqs, size=int(rng.integers(1, max_seqs + 1)), replace=False)).astype(np.int64)
    use_rows = seed % 2 == 0
    want = _numpy_build(live, num_tokens, n_draft_all, draft_ids, draft_row if use_rows else None, lstm_k, bt_host, bs, G)

    A, Bf = np.zeros(1 << 17, np.uint8), np.zeros(1 << 17, np.uint8)
    oa, ob, tot, cs = np.zeros(5, np.int64), np.zeros(7, np.int64), np.zeros(8, np.int64), np.zeros(1, np.int64)
    N.check(N.lib().aic_step_build(len(live), live.ctypes.data, max_seqs, num_tokens.ctypes.data, n_draft_all.ctypes.data,
                                   draft_ids.ctypes.data, MAX_SPEC_LEN, draft_row.ctypes.data if use_rows else None, lstm_k,
                                   bt_host.ctypes.data, bps, bs, G, A.ctypes.data, A.size, Bf.ctypes.data, Bf.size,
                                   oa.ctypes.data, ob.ctypes.data, tot.ctypes.data, cs.ctypes.data))
    T, max_q, max_ctx, n_short, D, F, bytes_a, bytes_b = (int(x) for x in tot)
    n = len(live)
    sec = lambda buf, off, cnt, dt: buf[off:off + cnt * np.dtype(dt).itemsize].view(dt)
    assert (T, max_q, max_ctx, n_short, int(cs[0])) == (want["T"], want["max_q"], want["max_ctx"], want["n_short"], want["ctx_sum"])
    assert D == len(want["draft_flat"]) and F == len(want["fill_pos"])
    assert all(o % 16 == 0 for o in list(oa) + list(ob)) and bytes_a <= A.size and bytes_b <= Bf.size
    assert np.array_equal(sec(A, oa[0], n, np.int32), want["ctx"])
    assert np.array_equal(sec(A, oa[1], n + 1, np.int32), want["qsl"])
    assert np.array_equal(sec(A, oa[2], n, np.int64), live)
    assert np.array_equal(sec(A, oa[3], T, np.int64), want["slots"])
    assert np.array_equal(sec(A, oa[4], n, np.int32), want["order"])
    assert np.array_equal(sec(Bf, ob[0], D, np.int32), want["draft_flat"])
    assert np.array_equal(sec(Bf, ob[1], n, np.int32), want["cu_draft"])
    assert np.array_equal(sec(Bf, ob[3], D, np.int64), want["target_rows"])
    assert np.array_equal(sec(Bf, ob[4], n, np.int64), want["bonus_rows"])
    assert np.array_equal(sec(Bf, ob[5], F, np.int64), want["fill_pos"])
    assert np.array_equal(sec(Bf, ob[6], F, np.int64), want["fill_src"])
    # the plant section is left to the caller and overlaps nothing
    assert ob[2] + 8 * T <= ob[3]


def test_step_build_refuses_small_buffers_and_bad_state():
    live = np.array([0], np.int64)
    nt, nd = np.array([5], np.int32), np.array([2], np.int32)
    ids, bt = np.zeros((1, MAX_SPEC_LEN), np.int32), np.zeros((1, 4), np.int32)
    A, B = np.zeros(16, np.uint8), np.zeros(1 << 12, np.uint8)
    o5, o7, t8, c1 = np.zeros(5, np.int64), np.zeros(7, np.int64), np.zeros(8, np.int64), np.zeros(1, np.int64)
    call = lambda a, bts, slots=1: N.lib().aic_step_build(1, live.ctypes.data, slots, nt.ctypes.data, nd.ctypes.data,
                                                          ids.ctypes.data, MAX_SPEC_LEN, None, 0, bt.ctypes.data, bts, 16, 4,
                                                          a.ctypes.data, a.size, B.ctypes.data, B.size, o5.ctypes.data,
                                                          o7.ctypes.data, t8.ctypes.data, c1.ctypes.data)
    # too small: a code of its own (the engine used to match the message text), and the needed sizes come back
    assert call(A, 4) == N.AIC_ERR_BUFFER_TOO_SMALL and b"too small" in N.lib().aic_last_error()
    need_a, need_b = int(t8[6]), int(t8[7])
    assert need_a > A.size and 0 < need_b <= B.size
    assert call(np.zeros(need_a, np.uint8), 4) == 0 and int(t8[6]) == need_a
    assert call(np.zeros(need_a - 1, np.uint8), 4) == N.AIC_ERR_BUFFER_TOO_SMALL
    nt[0] = 70                      # positions 69..71 need block 4 of a 4-block table
    assert call(np.zeros(1 << 12, np.uint8), 4) == N.AIC_ERR_INVALID and b"block table" in N.lib().aic_last_error()
    # a slot id outside the batch is refused before any per-slot array is read
    nt[0] = 5
    for bad in (1, -1, 1 << 40):
        live[0] = bad
        assert call(np.zeros(1 << 12, np.uint8), 4) == N.AIC_ERR_INVALID and b"not a slot" in N.lib().aic_last_error()
    live[0] = 0


@pytest.mark.parametrize("seed", range(4))
def test_step_parse_equals_the_numpy_expressions(seed):
    rng = np.random.default_rng(100 + seed)
    max_seqs, W, vocab, width = 16, 200, 5000, 9
    rows = rng.integers(0, vocab, (max_seqs, W)).astype(np.int32)
    num_tokens = rng.integers(1, W - width - 1, max_seqs).astype(np.int32)
    live = np.sort(rng.choice(max_seqs, size=10, replace=False)).astype(np.int64)
    out = np.full((len(live), width), -1, np.int32)
    for i in range(len(live)):
        k = int(rng.integers(1, width + 1))
        out[i, :k] = rng.integers(0, vocab + 40, k)        # a few ids >= vocab: dropped, like -1 (parse_output)
    want_rows, want_nt = rows.copy(), num_tokens.copy()
    valid = (out != -1) & (out < vocab)
    n_emit = valid.sum(axis=1).astype(np.int32)
    flat = out[valid]
    first = np.cumsum(n_emit) - n_emit
    within = np.arange(len(flat)) - np.repeat(first, n_emit)
    want_rows[np.repeat(live, n_emit), np.repeat(want_nt[live], n_emit) + within] = flat
    want_nt[live] += n_emit
    got_emit, got_flat, total = np.zeros(max_seqs, np.int32), np.zeros(max_seqs * width, np.int32), np.zeros(1, np.int64)
    N.check(N.lib().aic_step_parse(len(live), live.ctypes.data, max_seqs, out.ctypes.data, width, vocab, rows.ctypes.data, W,
                                   num_tokens.ctypes.data, got_emit.ctypes.data, got_flat.ctypes.data, total.ctypes.data))
    assert int(total[0]) == len(flat) and np.array_equal(got_flat[:len(flat)], flat)
    assert np.array_equal(got_emit[:len(live)], n_emit)
    assert np.array_equal(rows, want_rows) and np.array_equal(num_tokens, want_nt)


def test_step_parse_commits_nothing_when_a_later_row_is_bad():
    """an overflow (or a bad slot id) in row i used to leave rows 0..i-1 committed — num_tokens advanced for some
    slots only.  The validation pass runs over every row before the first write."""
    max_seqs, W, vocab, width = 4, 12, 100, 4
    rows = np.arange(max_seqs * W, dtype=np.int32).reshape(max_seqs, W) % vocab
    num_tokens = np.array([3, 5, 10, 2], np.int32)          # slot 2 has room for 2 more tokens only
    out = np.array([[7, 8, -1, -1], [9, -1, -1, -1], [1, 2, 3, -1], [4, -1, -1, -1]], np.int32)
    emit, flat, total = np.zeros(max_seqs, np.int32), np.zeros(max_seqs * width, np.int32), np.zeros(1, np.int64)

    def call(live):
        live = np.asarray(live, np.int64)
        return N.lib().aic_step_parse(len(live), live.ctypes.data, max_seqs, out.ctypes.data, width, vocab, rows.ctypes.data, W,
                                      num_tokens.ctypes.data, emit.ctypes.data, flat.ctypes.data, total.ctypes.data)

    r0, n0 = rows.copy(), num_tokens.copy()
    assert call([0, 1, 2, 3]) == N.AIC_ERR_INVALID and b"overflow" in N.lib().aic_last_error()
    assert np.array_equal(rows, r0) and np.array_equal(num_tokens, n0)
    for bad in (4, -1):
        assert call([0, 1, bad, 3]) == N.AIC_ERR_INVALID and b"not a slot" in N.lib().aic_last_error()
        assert np.array_equal(rows, r0) and np.array_equal(num_tokens, n0)
    num_tokens[2] = 9                                        # now it fits exactly
    assert call([0, 1, 2, 3]) == 0 and int(total[0]) == 7
    assert num_tokens.tolist() == [5, 6, 12, 3] and rows[2, 9:12].tolist() == [1, 2, 3]
