EMF_TRACK_WINDOW=0 EMF_TRACK_CHUNK=1 EMF_TRACK_LOG=1 FRAMES=3 python scripts/track_verdict_sequences.py 2>&1 | grep "track model" | head -150
