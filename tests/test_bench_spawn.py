"""CPU: `python bench.py --gpus N` launches its own ranks (the form the driver runs), and the same command under
torch.distributed.run uses the launcher's store -- both with --spawn-check, where every rank reports what it was given
and proves the rendezvous channel the RCCL id travels by, WITHOUT touching a GPU or loading the library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MJHMC_COMM_ID_FILE', 'MASTER_ADDR', 'MASTER_PORT',
              'TORCHELASTIC_USE_AGENT_STORE'):
        env.pop(k, None)
    return env


def test_bench_spawns_its_own_ranks():
    p = subprocess.run([sys.executable, BENCH, '--gpus', '3', '--steps', '20', '--warmup', '5', '--spawn-check'],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1                                  # ONE line on stdout
    rep = json.loads(lines[0])
    ranks = rep['spawn_check']
    assert rep['n_gpus'] == 3 and [r['rank'] for r in ranks] == [0, 1, 2] and [r['local_rank'] for r in ranks] == [0, 1, 2]
    assert all(r['world'] == 3 and r['id_file_explicit'] and r['channel'] == 'FileRendezvous' for r in ranks)
    assert len({r['id_file'] for r in ranks}) == 1 and 'mjhmc_bench_' in ranks[0]['id_file']
    assert all(r['probe'] == 'id-from-rank-0' and r['all_ranks_agree'] for r in ranks)
    assert not any(r['library_loaded'] for r in ranks)      # nothing touched the GPU library before the ranks existed
    assert not os.path.exists(os.path.dirname(ranks[0]['id_file']))      # the rendezvous directory is gone


def test_bench_under_torch_distributed_run_uses_the_launchers_store():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port), BENCH, '--gpus', '2', '--spawn-check'],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    reps = [json.loads(x) for x in p.stdout.decode().replace('}{', '}\n{').splitlines() if x.strip().startswith('{')]
    assert sorted(r['rank'] for r in reps) == [0, 1]
    assert all(r['channel'] == 'StoreRendezvous' and r['probe'] == 'id-from-rank-0' and r['all_ranks_agree'] for r in reps)


def test_a_failing_rank_fails_the_launcher():
    """No GPU here: every rank dies creating its context -> the parent must exit non-zero and print no JSON line."""
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip('GPU present')
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert p.stdout.decode().strip() == ''
    assert 'stopping the other ranks' in p.stderr.decode() or 'exited with status' in p.stderr.decode()


def test_world_size_mismatch_is_an_error_not_an_assert():
    env = dict(_clean_env(), WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and 'WORLD_SIZE=4' in p.stderr.decode() and 'AssertionError' not in p.stderr.decode()


def test_file_rendezvous_and_stale_proof_default_path(tmp_path, monkeypatch):
    from mjhmc_amd.parallel import FileRendezvous, default_id_path
    a, b = FileRendezvous(tmp_path, 0, 2, timeout=5), FileRendezvous(tmp_path, 1, 2, timeout=5)
    a.put('k', b'\x00\x01binary')
    assert b.get('k') == b'\x00\x01binary'
    b.put('ok.1', b'0')                                     # one rank says no -> nobody goes
    assert a.all_agree('ok', True) is False
    b.put('yes.1', b'1')
    assert a.all_agree('yes', True) is True
    import pytest
    with pytest.raises(RuntimeError):
        FileRendezvous(tmp_path, 1, 2, timeout=0.05).get('never')
    monkeypatch.delenv('MJHMC_COMM_ID_FILE', raising=False)
    path, explicit = default_id_path()
    # launcher pid AND its start time: a crashed earlier job with a recycled pid cannot have left this name behind
    assert not explicit and str(os.getppid()) in path and len(os.path.basename(path).split('_')) >= 5
    monkeypatch.setenv('MJHMC_COMM_ID_FILE', str(tmp_path / 'x.id'))
    assert default_id_path() == (str(tmp_path / 'x.id'), True)


def _fake_run(extra_env, n=3):
    import time
    env = dict(_clean_env(), MJHMC_BENCH_FAKE='1', **extra_env)
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, '--gpus', str(n), '--steps', '2', '--warmup', '1'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return p, time.time() - t0


def test_a_hung_rank_is_found_by_the_watchdog_and_the_run_switches_to_gloo():
    """The first 8-GPU run must fail fast: the sample all-gather check runs FIRST, and the spawning parent watches the
    ranks' milestones.  A rank that hangs where a broken collective would (MJHMC_BENCH_FAKE: no GPU, nothing loaded) is
    found within MJHMC_BENCH_GATHER_TIMEOUT, the attempt is stopped, and FRESH processes run on the gloo safety net --
    the line says why."""
    p, dt = _fake_run({'MJHMC_BENCH_FAKE_HANG': '1', 'MJHMC_BENCH_GATHER_TIMEOUT': '3'})
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    rep = json.loads(lines[0])
    assert rep['fake'] and 'did not finish within 3 s' in rep['comm_note'] and 'gloo' in rep['comm_note']
    assert 'ranks [1]' in p.stderr.decode()                  # the watchdog names the rank that did not get through
    assert dt < 60, dt                                       # found within the limit, not at the end of some global budget


def test_a_healthy_run_is_not_disturbed_by_the_watchdog_and_a_stall_later_on_is_found_too():
    p, dt = _fake_run({})
    assert p.returncode == 0 and json.loads(p.stdout.decode().strip())['comm_note'] is None
    # a rank that goes silent AFTER the gather check: the stall limit catches it
    p, dt = _fake_run({'MJHMC_BENCH_FAKE_WORK_S': '3600', 'MJHMC_BENCH_RCCL_TIMEOUT': '3'})
    assert p.returncode == 0 and dt < 60
    assert 'no rank reported a milestone for 3 s' in json.loads(p.stdout.decode().strip())['comm_note']
