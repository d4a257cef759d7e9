"""Serving surface of the drop-in: the worker / controller HTTP protocol of streammind/serve/
(`model_worker.py:85-397`, `controller.py:57-298`) -- same routes, same JSON, same `\\0`-delimited chunk stream -- in front of
the native model, plus the streaming-gate endpoint the reference lacks (`/worker_stream_frames`)."""
