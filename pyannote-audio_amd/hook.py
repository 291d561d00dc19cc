"""Pipeline hooks: the callables a user hands to `pipeline(file, hook=...)` -- same names, constructor
arguments, context-manager use and call protocol as pipelines/utils/hook.py:37-240 of the reference:

    hook(step_name, step_artifact, file=None, total=None, completed=None)

`Pipeline.setup_hook` (core/pipeline.py) binds `file`; a stage reports progress with
`total` / `completed` (artifact None) and its result with an artifact (no counters).  The reference's own
speed numbers come from `TimingHook`: with it the same measurement can be taken on this build."""
from __future__ import annotations

import copy
import time
from typing import Any, Mapping, Optional

import torch


class ArtifactHook:
    """Keeps a deep copy of every step's artifact in `file[file_key][step_name]`
    (all steps, or only the named ones)."""

    def __init__(self, *artifacts: str, file_key: str = "artifact"):
        self.artifacts = artifacts
        self.file_key = file_key

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return None

    def __call__(self, step_name: str, step_artifact: Any, file: Optional[Mapping] = None,
                 total: Optional[int] = None, completed: Optional[int] = None):
        wanted = not self.artifacts or step_name in self.artifacts
        if step_artifact is None or not wanted:
            return
        if isinstance(step_artifact, torch.Tensor):
            step_artifact = step_artifact.numpy(force=True)
        file.setdefault(self.file_key, {})[step_name] = copy.deepcopy(step_artifact)


class TimingHook:
    """Wall-clock seconds per step in `file[file_key]` (+ "total" between __enter__ and __exit__).
    A step is timed from its `completed == 0` report to its first `completed >= total` report."""

    def __init__(self, file_key: str = "timing"):
        self.file_key = file_key

    def __enter__(self):
        self._entered = time.time()
        self._begin: dict = {}
        self._finish: dict = {}
        return self

    def __exit__(self, *exc):
        now = time.time()
        report = {"total": now - self._entered}
        report.update({step: self._finish[step] - begun for step, begun in self._begin.items()})
        self._file[self.file_key] = report

    def __call__(self, step_name: str, step_artifact: Any, file: Optional[Mapping] = None,
                 total: Optional[int] = None, completed: Optional[int] = None):
        if not hasattr(self, "_file"):
            self._file = file
        if completed is None:
            return
        if completed == 0:
            self._begin[step_name] = time.time()
        if completed >= total:
            self._finish[step_name] = time.time()


class ProgressHook:
    """One progress bar per step (rich, like the reference); `hidden=True` turns it into a no-op,
    `transient=True` clears the display on exit."""

    def __init__(self, transient: bool = False, hidden: bool = False):
        self.transient = transient
        self.hidden = hidden

    def __enter__(self):
        if not self.hidden:
            from rich.progress import (BarColumn, Progress, TaskProgressColumn, TextColumn,
                                       TimeRemainingColumn)
            self.progress = Progress(TextColumn("[progress.description]{task.description}"), BarColumn(),
                                     TaskProgressColumn(), TimeRemainingColumn(elapsed_when_finished=True),
                                     transient=self.transient)
            self.progress.start()
        return self

    def __exit__(self, *exc):
        if not self.hidden:
            self.progress.stop()

    def __call__(self, step_name: str, step_artifact: Any, file: Optional[Mapping] = None,
                 total: Optional[int] = None, completed: Optional[int] = None):
        if self.hidden:
            return
        if completed is None:                      # an artifact report closes the step's bar
            completed = total = 1
        if getattr(self, "step_name", None) != step_name:
            self.step_name = step_name
            self.step = self.progress.add_task(step_name)
        self.progress.update(self.step, completed=completed, total=total)
        if completed >= total:
            self.progress.refresh()


class Hooks:
    """Several hooks as one: `with Hooks(ProgressHook(), TimingHook()) as hook: pipeline(file, hook=hook)`."""

    def __init__(self, *hooks):
        self.hooks = hooks

    def __enter__(self):
        for hook in self.hooks:
            if hasattr(hook, "__enter__"):
                hook.__enter__()
        return self

    def __exit__(self, *exc):
        for hook in self.hooks:
            if hasattr(hook, "__exit__"):
                hook.__exit__(*exc)

    def __call__(self, step_name: str, step_artifact: Any, file: Optional[Mapping] = None,
                 total: Optional[int] = None, completed: Optional[int] = None):
        for hook in self.hooks:
            hook(step_name, step_artifact, file=file, total=total, completed=completed)
