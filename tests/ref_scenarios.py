"""The reference's own rule tests (tests/cpp/game_test.cpp, move_test.cpp,
node_test.cpp) restated over an abstract backend, so the same known-answer
scenarios run against the CPU oracle (tests/test_oracle_reference_tests.py)
and against the HIP rule kernels through the C ABI (tests/test_engine_parity.py::test_reference_rule_scenarios,
`hip` and `emu` engines).

A backend provides:
  new_game() -> game with .legal_moves() -> (list[96] of bool, is_lines),
                .do_move(id), .state() -> float32[70], .copy(),
                .terminal_result() -> 0 none / 1 loss / 2 draw
  encode_place(row, col, piece), encode_move(r0, c0, r1, c1), decode_move(id)
"""
import numpy as np

kBase, kColumn, kCapital, kFrozen = 0, 1, 2, 3
PIECES = (kBase, kColumn, kCapital)


def sp(a, b, flip=False):
    """Space{a, b, flip} (util.h:24-33)"""
    return (b, a) if flip else (a, b)


def _place(B, space, piece):
    return B.encode_place(space[0], space[1], piece)


def _move(B, a, b):
    return B.encode_move(a[0], a[1], b[0], b[1])


# ---- game_test.cpp:6-26 and :28-64 (first half) ----
def default_constructor(B):
    g = B.new_game()
    lm, lines = g.legal_moves()
    assert not lines
    for r in range(4):
        for c in range(4):
            for p in PIECES:
                assert lm[B.encode_place(r, c, p)]
    for i in range(48):
        assert not lm[i]


# ---- game_test.cpp:28-64 (second half): terminal position with a line ----
def web_app_constructor(B):
    board = [0] * 64
    board[2 * 4 + kCapital] = 1
    board[5 * 4 + kCapital] = 1
    board[8 * 4 + kCapital] = 1
    board[8 * 4 + kFrozen] = 1
    g = B.game_from_arrays(board, 1, [4, 4, 2, 4, 4, 3])
    lm, lines = g.legal_moves()
    assert lines
    assert not any(lm)
    assert g.terminal_result() == 1  # loss for the side to move (node.cpp:261-266)


# ---- game_test.cpp:66-124 ----
def place_on_empty_board(B):
    for row in range(4):
        for col in range(4):
            for piece in PIECES:
                g = B.new_game()
                g.do_move(B.encode_place(row, col, piece))
                lm, lines = g.legal_moves()
                assert not lines
                for r2 in range(4):
                    for c2 in range(4):
                        for p2 in PIECES:
                            assert lm[B.encode_place(r2, c2, p2)] == (not (row == r2 and col == c2))
                assert not any(lm[:48])
                st = g.state()
                for i in range(64):
                    want = 1.0 if i in (row * 16 + col * 4 + piece, row * 16 + col * 4 + kFrozen) else 0.0
                    assert st[i] == want
                assert list(st[64:67]) == [1.0, 1.0, 1.0]
                for i in range(3):
                    assert st[67 + i] == (0.75 if i == piece else 1.0)


# ---- game_test.cpp:126-175 ----
def move_capital_on_base_column(B):
    for y in range(4):
        for x in range(3):
            g = B.new_game()
            g.do_move(B.encode_place(y, x, kBase))
            g.do_move(B.encode_place(y, x + 1, kCapital))
            g.do_move(B.encode_place(y, x, kColumn))
            g.do_move(B.encode_place((y + 1) % 4, x, kBase))
            g.do_move(B.encode_move(y, x + 1, y, x))
            lm, lines = g.legal_moves()
            assert not lines
            assert not any(lm[:48])
            for row in range(4):
                for col in range(4):
                    for p in PIECES:
                        bad = (row == y and col == x) or (row == (y + 1) % 4 and col == x and p in (kBase, kCapital))
                        assert lm[B.encode_place(row, col, p)] == (not bad)
            st = g.state()
            for i in range(64):
                want = 1.0 if (i // 4 == y * 4 + x or i == ((y + 1) % 4) * 16 + x * 4 + kBase) else 0.0
                assert st[i] == want


# ---- game_test.cpp:177-224 ----
def column_capital_on_base(B):
    for y in range(4):
        for x in range(3):
            g = B.new_game()
            g.do_move(B.encode_place(y, x, kColumn))
            g.do_move(B.encode_place(y, x + 1, kBase))
            g.do_move(B.encode_place(y, x, kCapital))
            g.do_move(B.encode_place((y + 1) % 4, x, kCapital))
            g.do_move(B.encode_move(y, x, y, x + 1))
            lm, lines = g.legal_moves()
            assert not lines
            assert not any(lm[:48])
            for row in range(4):
                for col in range(4):
                    for p in PIECES:
                        bad = (row == y and col == x + 1) or (row == (y + 1) % 4 and col == x)
                        assert lm[B.encode_place(row, col, p)] == (not bad)
            st = g.state()
            for i in range(64):
                want = 1.0 if (i // 4 == y * 4 + (x + 1) or i == ((y + 1) % 4) * 16 + x * 4 + kCapital) else 0.0
                assert st[i] == want


# ---- game_test.cpp:226-253 ----
def no_piece_left(B):
    for y in range(4):
        for x in range(4):
            g = B.new_game()
            for dy in (0, 1):
                g.do_move(B.encode_place((y + dy) % 4, x, kBase))
                g.do_move(B.encode_place((y + dy) % 4, (x + 1) % 4, kColumn))
                g.do_move(B.encode_place((y + dy) % 4, (x + 2) % 4, kBase))
                g.do_move(B.encode_place((y + dy) % 4, (x + 3) % 4, kCapital))
            lm, lines = g.legal_moves()
            assert not lines
            for row in range(4):
                for col in range(4):
                    assert not lm[B.encode_place(row, col, kBase)]


# ---- game_test.cpp:255-271 ----
def place_piece_on_same(B):
    for row in range(4):
        for col in range(4):
            for p in PIECES:
                g = B.new_game()
                g.do_move(B.encode_place(row, col, p))
                g.do_move(B.encode_place((row + 1) % 4, (col + 1) % 4, p))
                lm, lines = g.legal_moves()
                assert not lines
                assert not lm[B.encode_place(row, col, p)]


# ---- game_test.cpp:273-288 ----
def move_with_frozen(B):
    for y in range(4):
        for x in range(4):
            g = B.new_game()
            g.do_move(B.encode_place(y, (x + 3) % 4, kCapital))
            g.do_move(B.encode_place(y, (x + 1) % 4, kBase))
            g.do_move(B.encode_place(y, x, kColumn))
            lm, lines = g.legal_moves()
            assert not lines
            assert not any(lm[:48])


def _all_lines_broken_after(B, g, move_id):
    g2 = g.copy()
    g2.do_move(move_id)
    _, lines2 = g2.legal_moves()
    assert not lines2


# ---- game_test.cpp:290-346 ----
def long_row_cols(B):
    for flip in (False, True):
        for row in range(4):
            for p in PIECES:
                g = B.new_game()
                for k in range(3):
                    g.do_move(_place(B, sp(row, k, flip), p))
                lm, lines = g.legal_moves()
                assert lines
                assert sum(lm) == (1 if p == kCapital else 3)
                assert lm[_place(B, sp(row, 3, flip), p)]
                g.do_move(_place(B, sp(row, 3, flip), p))
                lm, lines = g.legal_moves()
                assert lines
                if p == kCapital:
                    assert not any(lm)
                    assert g.terminal_result() == 1
                else:
                    assert sum(lm) == 2
                    assert not any(lm[:48])
                    for r2 in range(4):
                        for c2 in range(4):
                            for p2 in PIECES:
                                mid = _place(B, sp(r2, c2, flip), p2)
                                if r2 == row and c2 in (1, 2):
                                    if lm[mid]:
                                        _all_lines_broken_after(B, g, mid)
                                else:
                                    assert not lm[mid]


# ---- game_test.cpp:348-405 ----
def short_row_cols(B):
    for flip in (False, True):
        for row in range(4):
            for p in (kBase, kColumn):
                g = B.new_game()
                g.do_move(_place(B, sp(row, 0, flip), kCapital))
                for k in (1, 2, 3):
                    g.do_move(_place(B, sp(row, k, flip), p))
                lm, lines = g.legal_moves()
                assert lines
                assert sum(lm) == (3 if p == kColumn else 2)
                if p == kBase:
                    assert not any(lm[:48])
                else:
                    for i in range(48):
                        if lm[i]:
                            _all_lines_broken_after(B, g, i)
                for r2 in range(4):
                    for c2 in range(4):
                        for p2 in PIECES:
                            mid = _place(B, sp(r2, c2, flip), p2)
                            if r2 == row and c2 > 0:
                                if lm[mid]:
                                    _all_lines_broken_after(B, g, mid)
                            else:
                                assert not lm[mid]


# ---- game_test.cpp:407-458 ----
def long_diags(B):
    for flip in (False, True):
        for p in PIECES:
            g = B.new_game()
            g.do_move(B.encode_place(0, 3 if flip else 0, p))
            g.do_move(B.encode_place(1, 2 if flip else 1, p))
            g.do_move(B.encode_place(2, 1 if flip else 2, p))
            lm, lines = g.legal_moves()
            assert lines
            assert sum(lm) == (1 if p == kCapital else 3)
            assert lm[B.encode_place(3, 0 if flip else 3, p)]
            g.do_move(B.encode_place(3, 0 if flip else 3, p))
            lm, lines = g.legal_moves()
            assert lines
            if p == kCapital:
                assert not any(lm)
            else:
                assert sum(lm) == 2
                assert not any(lm[:48])
                for row in range(4):
                    for col in range(4):
                        mid = _place(B, sp(row, col, flip), p)
                        if (row == 1 and col == (2 if flip else 1)) or (row == 2 and col == (1 if flip else 2)):
                            if lm[mid]:
                                _all_lines_broken_after(B, g, mid)
                        else:
                            assert not lm[mid]


# ---- game_test.cpp:460-506 ----
def short_diags(B):
    seqs = [[(2, 0), (1, 1), (0, 2)], [(0, 1), (1, 2), (2, 3)], [(1, 3), (2, 2), (3, 1)], [(3, 2), (2, 1), (1, 0)]]
    for p in PIECES:
        for seq in seqs:
            g = B.new_game()
            for s in seq:
                g.do_move(_place(B, s, p))
            lm, lines = g.legal_moves()
            assert lines
            if p == kCapital:
                assert not any(lm)
            else:
                assert sum(lm) == 2
                assert not any(lm[:48])
                for row in range(4):
                    for col in range(4):
                        mid = B.encode_place(row, col, p)
                        if (row, col) in seq:
                            if lm[mid]:
                                _all_lines_broken_after(B, g, mid)
                        else:
                            assert not lm[mid]


# ---- game_test.cpp:508-527 ----
def two_lines(B):
    for row in (1, 2):
        for col in (1, 2):
            for p in PIECES:
                g = B.new_game()
                g.do_move(B.encode_place(row - 1, col, p))
                g.do_move(B.encode_place(row + 1, col, p))
                g.do_move(B.encode_place(row, col - 1, p))
                g.do_move(B.encode_place(row, col + 1, p))
                g.do_move(B.encode_place(row, col, p))
                lm, lines = g.legal_moves()
                assert lines
                assert not any(lm)


# ---- node_test.cpp:29-47 ----
def node_terminal(B):
    for row in range(4):
        g = B.new_game()
        assert g.terminal_result() == 0
        for col in range(4):
            g.do_move(B.encode_place(row, col, kCapital))
            assert (g.terminal_result() != 0) == (col == 3)
        assert g.terminal_result() == 1


# ---- move_test.cpp ----
def move_codec(B):
    is_place, piece, r0, c0, r1, c1 = B.decode_move(0)
    assert (is_place, r0, c0, r1, c1) == (0, 0, 0, 0, 1)
    is_place, piece, r0, c0, r1, c1 = B.decode_move(48)
    assert (is_place, piece, r1, c1) == (1, kBase, 0, 0)
    assert B.encode_place(0, 0, kBase) == 48
    assert B.encode_place(2, 3, kColumn) == 75
    assert B.encode_place(3, 1, kCapital) == 93
    assert B.encode_move(0, 0, 0, 1) == 0
    assert B.encode_move(1, 2, 2, 2) == 18
    for i in range(48):
        is_place, piece, r0, c0, r1, c1 = B.decode_move(i)
        assert not is_place and B.encode_move(r0, c0, r1, c1) == i
    for i in range(48, 96):
        is_place, piece, r0, c0, r1, c1 = B.decode_move(i)
        assert is_place and B.encode_place(r1, c1, piece) == i


RULE_SCENARIOS = [
    default_constructor,
    web_app_constructor,
    place_on_empty_board,
    move_capital_on_base_column,
    column_capital_on_base,
    no_piece_left,
    place_piece_on_same,
    move_with_frozen,
    long_row_cols,
    short_row_cols,
    long_diags,
    short_diags,
    two_lines,
    node_terminal,
    move_codec,
]
