"""bench.py --gpus N run bare: the parent starts N rank processes itself (no torch.distributed.run), relays rank 0's JSON line and
propagates failures.  CPU-only: the ranks are stub workers, plus one real invocation that must fail with a clear message here
(no GPU in this container, and one device on the GPU box)."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench      # noqa: E402


def _stub(tmp_path, body):
    path = tmp_path / 'stub_worker.py'
    path.write_text(textwrap.dedent(body))
    return [sys.executable, str(path)]


def test_launcher_starts_ranks_and_relays_rank0_line(tmp_path, capfd):
    worker = _stub(tmp_path, '''
        import json, os, sys
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
        assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
        rdv = os.environ['GNN_BENCH_RDV']
        # the rendezvous file is shared by all ranks: rank 0 writes, the others wait for it
        if rank == 0:
            open(rdv + '.tmp', 'w').write('id'); os.replace(rdv + '.tmp', rdv)
            print('some progress text')
            print(json.dumps({'metric': 'stub', 'n_gpus': world, 'argv': sys.argv[1:]}))
        else:
            import time
            while not os.path.exists(rdv): time.sleep(0.01)
            print('rank', rank, 'chatter')
    ''')
    rc = bench.launch_ranks(3, ['--gpus', '3', '--steps', '2'], worker=worker)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = out.strip().splitlines()
    assert len(lines) == 1, out                       # exactly ONE JSON line on stdout
    assert json.loads(lines[0]) == {'metric': 'stub', 'n_gpus': 3, 'argv': ['--gpus', '3', '--steps', '2']}
    assert 'some progress text' in err and 'chatter' in err


def test_launcher_propagates_failure_and_stops_the_other_ranks(tmp_path, capfd):
    worker = _stub(tmp_path, '''
        import os, sys, time
        if os.environ['RANK'] == '1':
            time.sleep(0.3); sys.exit(7)
        time.sleep(60)
    ''')
    t0 = time.time()
    rc = bench.launch_ranks(2, [], worker=worker, grace_s=5.0)
    assert rc == 7 and time.time() - t0 < 20
    out, err = capfd.readouterr()
    assert out.strip() == '' and 'rank 1 of 2 exited with code 7' in err


def test_bare_multi_gpu_invocation_fails_clearly_without_the_devices():
    """`python bench.py --gpus 2` where fewer than 2 devices are visible: the children say so, the parent exits non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ''
    assert 'needs 2 devices' in r.stderr, r.stderr[-2000:]


def test_mismatched_world_size_is_rejected():
    env = dict(os.environ, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr


def test_watchdog_ends_a_rank_that_hangs(tmp_path):
    """A rank stuck in a collective must end by itself (exit 124), so that the launcher can stop the others: bench.Watchdog."""
    script = tmp_path / 'hang.py'
    script.write_text(textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        with bench.Watchdog(0.5, 'a collective that never returns', 3):
            time.sleep(30)
    """))
    t0 = time.time()
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 124 and time.time() - t0 < 20
    assert 'rank 3' in r.stderr and 'did not finish within' in r.stderr
