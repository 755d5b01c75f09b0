"""bench.py at N = 1 runs its measurement in a worker process and starts a
worker that DIES (the platform's "Memory access fault by GPU" ends a process
without a line) again; the line says so.  Host logic only: no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Result:
    def __init__(self, rc, out):
        self.returncode, self.stdout = rc, out


def run_supervisor(monkeypatch, capsys, script):
    import bench
    calls = []

    def fake(cmd, env=None, stdout=None):
        assert env["RT_BENCH_WORKER"] == "1" and stdout == subprocess.PIPE
        calls.append(cmd)
        return script[len(calls) - 1]
    monkeypatch.setattr(subprocess, "run", fake)
    rc = bench.supervise(["--gpus", "1", "--steps", "3"])
    return rc, calls, capsys.readouterr().out


def test_a_worker_that_succeeds_is_handed_on_verbatim(monkeypatch, capsys):
    line = b'{"metric": "ray-surface-ops/sec", "value": 1.5}\n'
    rc, calls, out = run_supervisor(monkeypatch, capsys, [Result(0, line)])
    assert rc == 0 and len(calls) == 1 and out.encode() == line
    assert calls[0][-4:] == ["--gpus", "1", "--steps", "3"]


def test_a_worker_that_dies_is_started_again(monkeypatch, capsys):
    line = b'banner\n{"metric": "m", "value": 2.0}\n'
    rc, calls, out = run_supervisor(monkeypatch, capsys, [
        Result(-6, b""), Result(-6, b"{"), Result(0, line)])
    assert rc == 0 and len(calls) == 3
    assert "--no-configs" not in calls[1] and calls[2][-1] == "--no-configs"
    d = json.loads(out)
    assert d["value"] == 2.0 and d["attempts"] == 3
    assert d["died"] == ["attempt 1: killed by signal 6",
                         "attempt 2: killed by signal 6"]


def test_an_error_the_worker_reports_is_not_retried(monkeypatch, capsys):
    rc, calls, out = run_supervisor(monkeypatch, capsys, [Result(1, b"")])
    assert rc == 1 and len(calls) == 1 and out == ""


def test_three_deaths_are_a_failure(monkeypatch, capsys):
    rc, calls, out = run_supervisor(monkeypatch, capsys,
                                    [Result(-6, b"")]*3)
    assert rc == 1 and len(calls) == 3 and out == ""


def test_a_worker_killed_after_its_line_has_measured(monkeypatch, capsys):
    line = b'{"metric": "m", "value": 3.0}\n'
    rc, calls, out = run_supervisor(monkeypatch, capsys, [Result(-11, line)])
    assert rc == 0 and len(calls) == 1
    d = json.loads(out)
    assert d["value"] == 3.0
    assert d["worker_killed_after_its_line_by_signal"] == 11
