#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
for round in 1 2; do
python3 tools/two_pipelines.py --mode single --batch 1024 --steps 40 2>/dev/null | tail -1
python3 tools/two_pipelines.py --mode plain --pipes 2 --batch 512 --steps 40 --offset-ms 0 2>/dev/null | tail -1
python3 tools/two_pipelines.py --mode plain --pipes 2 --batch 512 --steps 40 --offset-ms 4 2>/dev/null | tail -1
python3 tools/two_pipelines.py --mode plain --pipes 2 --batch 1024 --steps 20 --offset-ms 8 2>/dev/null | tail -1
python3 tools/two_pipelines.py --mode single --batch 512 --steps 80 2>/dev/null | tail -1
done
