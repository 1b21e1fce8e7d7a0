"""A frame's duration must not depend on the stream history of the process that embeds emf::EMFusion (VERDICT r03
weak #6 / next #7).  The frame overlaps the raycast (stream `main`) with the background's sweep (`aux`) and the list
rebuilds (`lists`); HIP maps the streams of one priority class onto a pool of four hardware queues, and with `main`
or `lists` in the class the application's own streams live in, 2, 5, 6 or 9 live foreign streams cost the frame
35-60 % (round 4, scripts/stream_history_probe.py).  The three streams now use the other two classes.
Every scenario runs in a process of its own; configs[1], frames 30-110.  Reference: one cv::cuda::Stream per object
in the host application, include/EMFusion/core/EMFusion.h:471."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _ms(scenario):
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "stream_history_probe.py"), scenario], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE_RESULT")]
    assert r.returncode == 0 and lines, r.stdout[-1000:] + r.stderr[-2000:]
    return float(lines[-1].split()[2])


def test_frame_time_does_not_depend_on_the_process_stream_history(dev):
    fresh = min(_ms("fresh"), _ms("fresh"))
    seen = {"fresh": fresh}
    # 8 create / destroy cycles + 8 live foreign streams (the verdict's case), a second instance in the process,
    # and the counts of foreign streams that hit the old stream classes hardest
    for scenario in ("both", "second", "foreign2", "foreign5", "foreign6", "foreign9"):
        seen[scenario] = _ms(scenario)
    worst = max(seen, key=seen.get)
    assert seen[worst] <= 1.10 * fresh, seen
    assert 0.3 < fresh < 1.5, seen
