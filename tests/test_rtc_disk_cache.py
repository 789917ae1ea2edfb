"""CPU: the on-disk code-object cache of the run-time specialised kernels (spcies_amd/csrc/code_cache.hpp), driven through the
library's test hook with a stand-in compiler - no GPU, no hiprtc.  Two PROCESSES that need the same program compile it once."""
import ctypes as C
import multiprocessing as mp
import os
import time

import pytest

from spcies_amd import _lib


def _call(text, work_ms=0, drop_memory=0):
    lib = _lib.load()
    src, chk = C.c_int(-1), C.c_ulonglong(0)
    rc = lib.spcies_hip_rtc_cache_selftest(text.encode(), work_ms, drop_memory, C.byref(src), C.byref(chk))
    return rc, src.value, chk.value


def _stats():
    lib = _lib.load()
    out = (C.c_long * 6)()
    assert lib.spcies_hip_rtc_cache_stats_ex(out, 6) == 0
    return dict(zip(("mem_hits", "disk_hits", "compiles", "evictions", "disk_writes", "disk_errors"), list(out)))


def _worker(cache_dir, text, work_ms, start_at, q):
    os.environ["SPCIES_HIP_CACHE_DIR"] = cache_dir
    os.environ.pop("SPCIES_HIP_DISK_CACHE", None)
    while time.time() < start_at:  # both processes ask at the same moment
        time.sleep(0.001)
    rc, src, chk = _call(text, work_ms)
    q.put((rc, src, chk, _stats()))


def test_two_processes_compile_once(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    start_at = time.time() + 3.0  # (spawned interpreters import the package first)
    text = "kernel text of one controller " * 50
    procs = [ctx.Process(target=_worker, args=(str(tmp_path), text, 1500, start_at, q)) for _ in range(3)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(rc == 0 for rc, *_ in got)
    sources = sorted(src for _, src, _, _ in got)
    assert sources == [1, 1, 2], sources  # one compiled (2), the others waited on the file lock and read the file (1)
    assert len({chk for _, _, chk, _ in got}) == 1
    assert sum(st["compiles"] for *_, st in got) == 1 and sum(st["disk_hits"] for *_, st in got) == 2
    files = sorted(os.listdir(tmp_path))
    assert sum(f.endswith(".hsaco") for f in files) == 1 and not any(f.startswith(".tmp-") for f in files)


def test_memory_then_disk_then_compile(tmp_path, monkeypatch):
    monkeypatch.setenv("SPCIES_HIP_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("SPCIES_HIP_DISK_CACHE", raising=False)
    text = f"program {os.getpid()} {time.time()}"
    s0 = _stats()
    rc, src, chk = _call(text, 0, drop_memory=1)
    assert (rc, src) == (0, 2)
    rc, src, chk2 = _call(text)
    assert (rc, src, chk2) == (0, 0, chk)  # memory
    rc, src, chk3 = _call(text, 0, drop_memory=1)
    assert (rc, src, chk3) == (0, 1, chk)  # a fresh process' view: the file
    s1 = _stats()
    assert s1["compiles"] - s0["compiles"] == 1 and s1["disk_hits"] - s0["disk_hits"] == 1 and s1["mem_hits"] - s0["mem_hits"] == 1
    assert s1["disk_writes"] - s0["disk_writes"] == 1
    # another text is another program
    rc, src, chk4 = _call(text + " ", 0)
    assert (rc, src) == (0, 2) and chk4 != chk


def test_corrupt_or_foreign_file_is_recompiled(tmp_path, monkeypatch):
    monkeypatch.setenv("SPCIES_HIP_CACHE_DIR", str(tmp_path))
    text = f"program to corrupt {os.getpid()}"
    rc, src, chk = _call(text, 0, drop_memory=1)
    assert (rc, src) == (0, 2)
    (path,) = [os.path.join(tmp_path, f) for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0x55  # one flipped bit in the code
    open(path, "wb").write(bytes(blob))
    rc, src, chk2 = _call(text, 0, drop_memory=1)
    assert (rc, src, chk2) == (0, 2, chk)  # the checksum does not match: compiled again, file replaced
    open(path, "wb").write(b"short")
    rc, src, chk3 = _call(text, 0, drop_memory=1)
    assert (rc, src, chk3) == (0, 2, chk)
    rc, src, _ = _call(text, 0, drop_memory=1)
    assert (rc, src) == (0, 1)


def test_disk_cache_off_and_unusable_directory(tmp_path, monkeypatch):
    text = f"no disk {os.getpid()}"
    monkeypatch.setenv("SPCIES_HIP_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("SPCIES_HIP_DISK_CACHE", "0")
    assert _call(text, 0, drop_memory=1)[:2] == (0, 2)
    assert _call(text, 0, drop_memory=1)[:2] == (0, 2) and os.listdir(tmp_path) == []
    monkeypatch.delenv("SPCIES_HIP_DISK_CACHE")
    monkeypatch.setenv("SPCIES_HIP_CACHE_DIR", "/proc/spcies_hip_cannot_exist")
    assert _call(text, 0, drop_memory=1)[:2] == (0, 2)  # silently without the disk


def test_failed_compilation_is_not_cached(tmp_path, monkeypatch):
    monkeypatch.setenv("SPCIES_HIP_CACHE_DIR", str(tmp_path))
    text = f"fails first {os.getpid()}"
    rc, _, _ = _call(text, -1, drop_memory=1)
    assert rc == -3 and b"told to fail" in _lib.load().spcies_hip_last_error()
    assert _call(text, 0)[:2] == (0, 2)


def test_memory_cache_is_bounded(tmp_path, monkeypatch):
    monkeypatch.setenv("SPCIES_HIP_DISK_CACHE", "0")
    monkeypatch.setenv("SPCIES_HIP_RTC_CACHE_MB", "0")  # cap 0: only the newest entry stays
    _call("bounded a", 0, drop_memory=1)
    e0 = _stats()["evictions"]
    _call("bounded b", 0)
    _call("bounded c", 0)
    assert _stats()["evictions"] - e0 == 2
    assert _call("bounded c", 0)[1] == 0 and _call("bounded a", 0)[1] == 2


def test_disk_cache_is_capped_and_leaves_no_lock_files(tmp_path, monkeypatch):
    """SPCIES_HIP_DISK_CACHE_MB: the directory does not grow without bound (round-4 advisor finding) - the least recently USED code
    objects go first - and a digest's lock file is gone once its code object is written."""
    monkeypatch.setenv("SPCIES_HIP_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("SPCIES_HIP_DISK_CACHE", raising=False)
    monkeypatch.setenv("SPCIES_HIP_DISK_CACHE_MB", "1")
    tag = f"{os.getpid()} {time.time()}"
    big = "x" * 300_000  # the stand-in compiler's code object grows with the text
    texts = [f"prune {tag} {i} {big}" for i in range(6)]
    rc, src, chk0 = _call(texts[0], 0, 1)
    assert rc == 0 and src == 2
    size = max(os.path.getsize(tmp_path / f) for f in os.listdir(tmp_path) if f.endswith(".hsaco"))
    if size * 6 < (1 << 20):
        pytest.skip(f"stand-in code objects are {size} B: six of them stay under the 1 MB cap")
    time.sleep(1.1)  # mtime resolution
    for t in texts[1:3]:
        assert _call(t, 0, 1)[0] == 0
        time.sleep(0.02)
    time.sleep(1.1)
    assert _call(texts[0], 0, 1)[1] == 1  # a disk hit: text 0 is now the most recently used
    time.sleep(1.1)
    assert _call(texts[3], 0, 1)[0] == 0  # the fourth file crosses the cap: the two least recently used (1, 2) go, 80 % of the cap is left
    files = os.listdir(tmp_path)
    total = sum(os.path.getsize(tmp_path / f) for f in files if f.endswith(".hsaco"))
    assert total <= (1 << 20), total
    assert not [f for f in files if f.endswith(".lock") and f != ".prune.lock"], files
    rc, src, chk = _call(texts[0], 0, 1)
    assert (rc, src, chk) == (0, 1, chk0)  # the recently used one survived the pruning
    assert _call(texts[1], 0, 1)[1] == 2   # the oldest one did not: compiled again


def test_a_program_that_killed_the_compiler_is_not_tried_again(tmp_path, monkeypatch):
    """Round 5: hiprtc runs in a helper process, and a program that kills it (ROCm 7.2 has such kernels) fails ONE build.  The disk cache marks
    the digest (<digest>.crashed), so the next process - every later MATLAB session - skips the attempt instead of crashing the helper again."""
    monkeypatch.setenv("SPCIES_HIP_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("SPCIES_HIP_DISK_CACHE", raising=False)
    text = f"crasher {os.getpid()} {time.time()}"
    rc1, _, _ = _call(text, -2, 1)
    assert rc1 == -7001 and [f for f in os.listdir(tmp_path) if f.endswith(".crashed")]
    s0 = _stats()
    rc2, _, _ = _call(text, 0, 1)  # (work_ms = 0 would compile fine: the marker answers before the compiler is asked)
    assert rc2 == -7002 and _stats()["compiles"] == s0["compiles"]
    for f in os.listdir(tmp_path):
        if f.endswith(".crashed"):
            os.unlink(tmp_path / f)
    rc3, src, _ = _call(text, 0, 1)
    assert rc3 == 0 and src == 2
