"""The transducer KERNELS on the CPU tier: pg_fsm_kernels.h's own source, compiled for the host and run by a thread-per-lane emulator.

tools/simt_emu/hip/hip_runtime.h stands in for the HIP runtime header (every lane an OS thread, __syncthreads / wave barriers / shuffles as
pthread barriers, V_PERM_B32 and V_BFE_U32 by their definitions); tools/simt_emu/fsm_emu_driver.cpp launches the kernels in the order
pg_engine.hip's device_fsm_filter_stats does -- tile walk (table / byte-function <= 4 states / <= 8 states), fsm_chain, fsm_finish, and for
a machine with a NOT child fsm_chunk_states, fsm_tile_states, fsm_episode_tiles, fsm_episode_finish.  The count must equal the oracle's
iterator objects.  What this buys: the index arithmetic, the barriers, the tails of the last tile and the chunk boundary (more than 1024
tiles) of the code the GPU runs are checked without a GPU -- round 5's last change to fsm_episode_finish_kernel was verified at 6 006 tiles
this way (12.3 M docs: 25 110 578 entries = the oracle; 74 s, not part of this file) when the round's GPU minutes were gone.
The host twins of tests/test_filter_stats_cpu.py restate the arithmetic; this runs the source."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle
from pinot_amd import _abi
from pinot_amd import query as Q
from pinot_amd import segment as S
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    out_dir = str(tmp_path_factory.mktemp("simt_emu"))
    src = open(os.path.join(ROOT, "pinot_amd", "csrc", "pg_fsm_kernels.h")).read()
    assert "extern __shared__" in src and "fsm_episode_finish_kernel" in src
    # a workgroup at a time: a function-local static IS the workgroup's LDS; the dynamic LDS array is defined by the driver
    with open(os.path.join(out_dir, "pg_fsm_kernels_emu.h"), "w") as f:
        f.write(src.replace("extern __shared__", "extern").replace("__shared__", "static"))
    lib_path = os.path.join(out_dir, "libfsm_emu.so")
    # -fvisibility=hidden / -Bsymbolic: the kernel templates are weak symbols, and libpinot_gpu.so -- which an earlier test of the same process may
    # have loaded -- holds HOST STUBS of the same mangled names (the kernel handles hipLaunchKernel takes): a call bound to one of those jumps
    # into a handle.  Nothing of this library is visible or interposable but emu_fsm_count.
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-pthread", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wl,-Bsymbolic", "-Wall", "-Wno-unknown-pragmas",
           "-I", os.path.join(ROOT, "tools", "simt_emu"), "-I", out_dir,
           "-I", os.path.join(ROOT, "include"), "-o", lib_path, os.path.join(ROOT, "tools", "simt_emu", "fsm_emu_driver.cpp")]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    lib = C.CDLL(lib_path)
    lib.emu_fsm_count.restype = C.c_int64
    lib.emu_fsm_count.argtypes = [C.POINTER(_abi.pg_query), C.c_int32, C.POINTER(C.POINTER(C.c_uint64)), C.c_int32, C.c_int32] + [C.POINTER(C.c_int32)] * 3
    return lib


def leaf_bitmaps(seg, spec):
    preds = spec.predicates
    keep, ptrs = [], (C.POINTER(C.c_uint64) * max(len(preds), 1))()
    for i, p in enumerate(preds):
        words, _ = oracle.filter_bitmap(seg, Q.QuerySpec([], filter=Q.leaf(p)))
        words = np.ascontiguousarray(np.concatenate([words, np.zeros(1, dtype=np.uint64)]))
        keep.append(words)
        ptrs[i] = words.ctypes.data_as(C.POINTER(C.c_uint64))
    return keep, ptrs


def run(emu, seg, spec, walk, blocks=3):
    keep, ptrs = leaf_bitmaps(seg, spec)
    states, inputs, episodes = C.c_int32(), C.c_int32(), C.c_int32()
    got = int(emu.emu_fsm_count(C.byref(spec.c), seg.num_docs, ptrs, walk, blocks, C.byref(states), C.byref(inputs), C.byref(episodes)))
    return got, int(states.value), int(inputs.value), bool(episodes.value)


def segment(rng, n):
    return S.SegmentData("emu_%d" % n, n, [H.random_dict_column(rng, "a", n, 50)[0], H.random_dict_column(rng, "d", n, 3)[0], H.random_dict_column(rng, "f", n, 2000)[0],
                                            H.random_dict_column(rng, "b", n, 7, with_inverted=True)[0], H.random_dict_column(rng, "e", n, 11)[0]])


def leaves():
    a, d, rare = Q.leaf(Q.Pred.dict_range(0, 3, 20)), Q.leaf(Q.Pred.dict_range(1, 1, 2)), Q.leaf(Q.Pred.dict_range(2, 100, 103))
    post, e = Q.leaf(Q.Pred.dict_range(3, 0, 3, inverted=True)), Q.leaf(Q.Pred.dict_range(4, 2, 6))
    return a, d, rare, post, e


@pytest.mark.parametrize("n", [1, 33, 2049, 70_003])
def test_the_kernels_source_equals_the_oracle(emu, n):
    """Named shapes through every walk that takes them: three scan leaves, an OR beside a scan leaf, the merged-bitmap form, NOT over a dense
    and over a rarely matching scan leaf (leading and not), NOT over an index-based leaf, five leaves, two NOT children over scan leaves (an episode stream each)."""
    rng = np.random.default_rng(100 + n)
    seg = segment(rng, n)
    a, d, rare, post, e = leaves()
    shapes = [Q.and_(a, d, rare), Q.and_(a, Q.or_(d, rare)), Q.and_(post, a, Q.or_(d, rare)), Q.and_(a, Q.not_(d)), Q.and_(a, Q.not_(rare)), Q.and_(Q.not_(rare), d, a),
              Q.and_(post, Q.not_(rare)), Q.and_(a, Q.or_(d, rare), Q.not_(post)), Q.and_(a, d, e, rare, Q.leaf(Q.Pred.dict_range(0, 5, 40))), Q.and_(e, Q.not_(d), Q.or_(a, post)),
              # two NOT children over scan leaves: two episode streams of one machine (7 states: the range kernel; 15: the table walk)
              Q.and_(Q.not_(a), Q.not_(rare)), Q.and_(a, Q.not_(d), Q.not_(rare)), Q.and_(post, Q.not_(rare), Q.not_(e)),
              # NOT over an OR of leaves (round 6c): an episode stream per scan member; 9 .. 16 states: fsm_tile_fns16_kernel + fsm_episode_ranges_kernel<16, 4> (walk 3)
              Q.and_(a, Q.not_(Q.or_(d, rare))), Q.and_(Q.not_(Q.or_(rare, e)), a), Q.and_(post, Q.not_(Q.or_(d, rare))), Q.and_(a, Q.not_(Q.or_(post, rare))),
              Q.and_(e, Q.not_(d), Q.not_(rare)), Q.and_(Q.not_(Q.or_(a, e)), rare), Q.and_(d, Q.not_(Q.or_(rare, e)), post),
              # nine to sixteen states WITHOUT episodes (the same predicates behind several leaves): walk 3 runs the range kernel as the counter
              Q.and_(Q.or_(rare, d, a), d, Q.or_(rare, d, a)), Q.and_(Q.or_(rare, a), Q.or_(d, rare), a, Q.or_(rare, d)), Q.and_(Q.or_(d, a, rare), Q.or_(rare, d), e, a)]
    ran = with_episodes = sixteen = 0
    for flt in shapes:
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=flt)
        want = oracle.execute(seg, spec).stats[1]
        for walk in (0, 1, 2, 3):
            got, states, inputs, episodes = run(emu, seg, spec, walk)
            if got == -2:
                assert walk > 0 and ((states > (4 if walk == 1 else 8) or inputs > 4) if walk < 3 else (states <= 8 or inputs > 4))
                continue
            assert got == want, (n, walk, states, inputs, episodes, got, want)
            ran += 1
            with_episodes += 1 if episodes else 0
            sixteen += 1 if walk == 3 else 0
    assert ran >= 24 and with_episodes >= 14 and sixteen >= 4, (ran, with_episodes, sixteen)


def test_more_than_one_chunk_of_tiles(emu):
    """2 200 013 docs = 1 075 tiles: fsm_chain_kernel's second chunk, fsm_chunk_states / fsm_tile_states across the chunk boundary, the finish
    kernel's sixteen ranges of 64-tile groups with a carry -- the byte-function walk and the table walk of an eight-state machine."""
    rng = np.random.default_rng(7)
    n = 2_200_013
    seg = segment(rng, n)
    a, d, rare, post, e = leaves()
    for flt, walk in ((Q.and_(a, Q.not_(rare)), 1), (Q.and_(Q.not_(d), e, a), 0), (Q.and_(a, Q.not_(Q.or_(d, rare))), 3)):
        spec = Q.QuerySpec([(Q.COUNT, -1)], filter=flt)
        want = oracle.execute(seg, spec).stats[1]
        got, states, inputs, episodes = run(emu, seg, spec, walk, blocks=4)
        assert episodes and got == want, (walk, states, inputs, got, want)
