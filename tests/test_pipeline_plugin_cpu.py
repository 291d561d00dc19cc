"""Plugin boundary b1: the behaviours the reference pins in its own
tests/test_pipeline_subfolder.py -- `$model/...` expansion, config location via `subfolder`, nested
pipelines loaded from checkpoint dicts, argument validation -- checked on OUR `Pipeline` with dummy
pipeline classes resolved by dotted name exactly as the reference resolves plugin classes
(core/pipeline.py:271-278: `token` and `cache_dir` are always injected into the constructor)."""
import os

import numpy as np

import pytest
import yaml

from pyannote_audio_amd.pipeline import Pipeline, expand_subfolders


class LeafPipeline(Pipeline):
    def __init__(self, token=None, cache_dir=None, **params):
        super().__init__()
        self.params_seen = dict(params, token=token, cache_dir=cache_dir)

    def apply(self, file, **kwargs):
        return None


class ParentPipeline(Pipeline):
    """owns one sub-pipeline given as an expanded checkpoint dict"""

    def __init__(self, sub=None, token=None, cache_dir=None):
        super().__init__()
        if isinstance(sub, dict):
            self.sub = Pipeline.from_pretrained(**sub)   # registered in _pipelines by __setattr__

    def apply(self, file, **kwargs):
        return None


LEAF = f"{LeafPipeline.__module__}.{LeafPipeline.__qualname__}"
PARENT = f"{ParentPipeline.__module__}.{ParentPipeline.__qualname__}"


def write(root, config, subfolder=None):
    d = os.path.join(str(root), subfolder) if subfolder else str(root)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "config.yaml")
    with open(path, "w") as fp:
        yaml.safe_dump(config, fp)
    return path


# ---------------------------------------------------------------- expand_subfolders
def test_expansion_record_fields():
    cfg = {"plain": "value", "nested": {"k": 1}, "emb": "$model/embeddings"}
    expand_subfolders(cfg, model_id="org/repo", token="tok")
    assert cfg["plain"] == "value" and cfg["nested"] == {"k": 1}
    assert cfg["emb"] == {"checkpoint": "org/repo", "revision": None, "subfolder": "embeddings",
                          "token": "tok", "cache_dir": None}


@pytest.mark.parametrize("ref,parent_sub,parent_rev,want_sub,want_rev", [
    ("$model/seg", "v1", None, "v1/seg", None),
    ("$model/weights@abc123", None, "main", "weights", "abc123"),
    ("$model/weights", None, "v2", "weights", "v2"),
    ("$model/weights@pinned", "folder", "ignored", "folder/weights", "pinned"),
    ("$model/emb", "a/b", None, "a/b/emb", None),
])
def test_expansion_subfolder_and_revision_rules(ref, parent_sub, parent_rev, want_sub, want_rev):
    cfg = {"m": ref}
    expand_subfolders(cfg, model_id="org/repo", parent_subfolder=parent_sub, parent_revision=parent_rev)
    assert cfg["m"]["subfolder"] == want_sub and cfg["m"]["revision"] == want_rev


def test_expansion_recurses_into_lists_and_dicts():
    cfg = {"items": [{"model": "$model/a"}, {"model": "$model/b"}], "outer": {"inner": "$model/deep"},
           "lst": ["$model/x", "plain"]}
    expand_subfolders(cfg, model_id="org/repo", parent_subfolder="p")
    assert [i["model"]["subfolder"] for i in cfg["items"]] == ["p/a", "p/b"]
    assert cfg["outer"]["inner"]["subfolder"] == "p/deep"
    assert cfg["lst"][0]["subfolder"] == "p/x" and cfg["lst"][1] == "plain"
    top = ["$model/part_a", "$model/part_b"]
    expand_subfolders(top, model_id="org/repo")
    assert [t["subfolder"] for t in top] == ["part_a", "part_b"]


def test_expansion_without_model_id():
    cfg = {"seg": "$model/seg"}
    expand_subfolders(cfg, model_id=None)
    assert cfg["seg"]["checkpoint"] is None


# ---------------------------------------------------------------- from_pretrained on local layouts
def test_config_location(tmp_path):
    write(tmp_path, {"pipeline": {"name": LEAF}})
    write(tmp_path, {"pipeline": {"name": LEAF, "params": {"marker": "sub"}}}, subfolder="sub")
    write(tmp_path, {"pipeline": {"name": LEAF, "params": {"marker": "deep"}}}, subfolder="a/b")
    root = Pipeline.from_pretrained(str(tmp_path))
    assert isinstance(root, LeafPipeline) and "marker" not in root.params_seen
    assert Pipeline.from_pretrained(str(tmp_path), subfolder="sub").params_seen["marker"] == "sub"
    assert Pipeline.from_pretrained(str(tmp_path), subfolder="a/b").params_seen["marker"] == "deep"
    with pytest.raises(Exception):
        Pipeline.from_pretrained(str(tmp_path), subfolder="missing")
    # token / cache_dir are always passed to the plugin class
    p = Pipeline.from_pretrained(str(tmp_path), token="T", cache_dir="/c")
    assert p.params_seen["token"] == "T" and p.params_seen["cache_dir"] == "/c"


def test_argument_validation(tmp_path):
    cfg_path = write(tmp_path, {"pipeline": {"name": LEAF}})
    with pytest.raises(ValueError, match="[Rr]evision"):
        Pipeline.from_pretrained(str(tmp_path), revision="main")
    with pytest.raises(ValueError, match="[Ss]ubfolder"):
        Pipeline.from_pretrained(cfg_path, subfolder="v1")
    with pytest.raises(ValueError, match="[Ss]ubfolder"):
        Pipeline.from_pretrained({"pipeline": {"name": LEAF}}, subfolder="v1")
    assert isinstance(Pipeline.from_pretrained(cfg_path), LeafPipeline)


def test_three_level_nesting(tmp_path):
    write(tmp_path, {"pipeline": {"name": PARENT, "params": {"sub": "$model/child"}}}, subfolder="v1")
    write(tmp_path, {"pipeline": {"name": PARENT, "params": {"sub": "$model/grandchild"}}},
          subfolder="v1/child")
    write(tmp_path, {"pipeline": {"name": LEAF}}, subfolder="v1/child/grandchild")
    parent = Pipeline.from_pretrained(str(tmp_path), subfolder="v1")
    assert isinstance(parent, ParentPipeline) and "sub" in parent._pipelines
    assert isinstance(parent.sub, ParentPipeline) and "sub" in parent.sub._pipelines
    assert isinstance(parent.sub.sub, LeafPipeline)


def test_one_level_nesting_from_root_and_from_dict(tmp_path, monkeypatch):
    write(tmp_path, {"pipeline": {"name": PARENT, "params": {"sub": "$model/child"}}})
    write(tmp_path, {"pipeline": {"name": LEAF}}, subfolder="child")
    parent = Pipeline.from_pretrained(str(tmp_path))
    assert isinstance(parent, ParentPipeline) and isinstance(parent.sub, LeafPipeline)
    # a config DICT resolves $model/... relative to the working directory
    monkeypatch.chdir(tmp_path)
    write(tmp_path, {"pipeline": {"name": LEAF}}, subfolder="seg")
    p = Pipeline.from_pretrained({"pipeline": {"name": PARENT, "params": {"sub": "$model/seg"}}})
    assert isinstance(p, ParentPipeline) and isinstance(p.sub, LeafPipeline)


def test_list_call_contract(tmp_path):
    """pipeline([files]) -> iterator of (file, output); duplicate URIs are refused (core/pipeline.py:
    570-578)."""
    import torch
    write(tmp_path, {"pipeline": {"name": LEAF}})
    p = Pipeline.from_pretrained(str(tmp_path))
    wav = torch.zeros(1, 16000)
    files = [{"waveform": wav, "sample_rate": 16000, "uri": u} for u in ("a", "b")]
    out = list(p(files))
    assert [f["uri"] for f, _ in out] == ["a", "b"] and all(o is None for _, o in out)
    with pytest.raises(ValueError, match="distinct URIs"):
        list(p([files[0], dict(files[0])]))


def test_speaker_bounds_contract(pipeline_dir):
    """speaker_diarization.py:565-590: unknown keyword arguments warn, a clustering that needs the number of
    speakers takes it from the file's reference annotation (bounds untouched) or raises."""
    import pyannote_audio_amd as pa
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir)
    assert pipeline._speaker_bounds(None, None, None, {}) == (None, 1, np.inf)
    assert pipeline._speaker_bounds(None, 2, 2, {}) == (2, 2, 2)
    with pytest.warns(UserWarning, match="Ignoring unexpected keyword arguments: foo"):
        pipeline._speaker_bounds(3, None, None, {"foo": 1})
    with pytest.raises(ValueError):
        pipeline._speaker_bounds(None, 3, 2, {})
    pipeline._expects_num_speakers = True                      # what KMeansClustering declares
    with pytest.raises(ValueError, match="num_speakers must be provided"):
        pipeline._speaker_bounds(None, None, None, {}, file={"uri": "x"})
    reference = pa.Annotation(uri="x")
    reference[pa.Segment(0, 1), "a"] = "alice"
    reference[pa.Segment(1, 2), "b"] = "bob"
    assert pipeline._speaker_bounds(None, None, None, {}, file={"uri": "x", "annotation": reference}) == (2, 1, np.inf)


def test_label_names(pipeline_dir):
    """speaker_diarization.py:716-737: SPEAKER_xx in labels() order, or the reference speakers when the file
    carries its annotation (hypothesis speakers without a counterpart keep their cluster id)."""
    import pyannote_audio_amd as pa
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir)
    hyp = pa.Annotation(uri="h")
    hyp[pa.Segment(0, 5), "_"] = 0
    hyp[pa.Segment(5, 9), "_"] = 1
    hyp[pa.Segment(20, 21), "_"] = 2
    assert pipeline._label_names({"uri": "h"}, hyp, hyp.labels()) == {0: "SPEAKER_00", 1: "SPEAKER_01", 2: "SPEAKER_02"}
    ref = pa.Annotation(uri="h")
    ref[pa.Segment(0, 5), "_"] = "alice"
    ref[pa.Segment(5, 10), "_"] = "bob"
    assert pipeline._label_names({"uri": "h", "annotation": ref}, hyp, hyp.labels()) == {0: "alice", 1: "bob", 2: 2}
    assert pipeline._label_names({"uri": "h", "annotation": pa.Annotation()}, hyp, hyp.labels())[0] == "SPEAKER_00"
