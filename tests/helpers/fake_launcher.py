"""Stand-in for `python -m torch.distributed.run` in tests/test_bench_spawn.py: called as
`fake_launcher.py [--lie K] N bench.py <bench args>`, it checks what bench.py's spawn logic hands to a launcher and
prints the JSON line rank 0 of an N-rank run would print (no GPU, no torch.distributed)."""
import json
import os
import sys

argv = sys.argv[1:]
lie = None
if argv[0] == "--lie":
    lie, argv = int(argv[1]), argv[2:]
n, bench, rest = int(argv[0]), argv[1], argv[2:]
assert os.path.basename(bench) == "bench.py", bench
assert "WORLD_SIZE" not in os.environ          # the parent must not have been a rank itself
assert rest[rest.index("--gpus") + 1] == str(n), rest
print("noise on stdout that is not the record")
print(json.dumps({"metric": "fake", "value": 1.0, "n_gpus": lie if lie is not None else n, "argv": rest}))
