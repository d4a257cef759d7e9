"""Callers of the streaming path (the reference's `streammind/eval/` scripts): the build's own harness."""
