"""tools/verify_checkpoint.py: the parity table for a user's own checkpoints.  Here: its CPU
half on the seeded checkpoints -- the real-format files load STRICTLY into the reference-shaped modules, the config is
resolved the way `Pipeline.from_pretrained` resolves it, the report is well formed.  The full table (HIP vs CPU) runs
in tests/test_pipeline_gpu.py::test_verify_checkpoint_tool."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool():
    spec = importlib.util.spec_from_file_location("verify_checkpoint", os.path.join(ROOT, "tools", "verify_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_half_on_seeded_checkpoints(tmp_path, capsys):
    from conftest import write_pipeline_dir
    from oracle.models import seeded_pyannet, seeded_wespeaker
    write_pipeline_dir(tmp_path, seeded_pyannet(), seeded_wespeaker())
    tool = load_tool()
    out_json = str(tmp_path / "report.json")
    rc = tool.main([str(tmp_path), os.path.join(ROOT, "tests", "golden", "sample.wav"), "--max-seconds", "11",
                    "--chunks", "1", "--oracle-only", "--json", out_json])
    text = capsys.readouterr().out
    assert rc == 0 and "strict state-dict match: ok" in text and "CPU side only" in text
    rep = json.load(open(out_json))
    assert rep["ok"] and rep["segmentation"] == "PyanNet" and rep["embedding"] == "WeSpeakerResNet34"
    stages = {(r["stage"], r["what"][:11]) for r in rep["rows"]}
    assert ("segmentation", "float32 CPU") in stages and ("embedding", "float32 CPU") in stages
    assert any(r["stage"] == "pipeline" for r in rep["rows"])
    # float32 against float64 on the reference's own arithmetic: far inside the tolerance on this audio
    assert all(r["value"] < 1.0 for r in rep["rows"] if r["what"].startswith("float32 CPU vs float64"))


def test_a_checkpoint_that_does_not_fit_the_class_is_reported(tmp_path):
    """strict loading: a missing key is an error of the tool, not a silently random layer"""
    from conftest import WESPEAKER_HPARAMS, write_pipeline_dir
    from oracle.models import seeded_pyannet, seeded_wespeaker
    from pyannote_audio_amd import model as pm
    write_pipeline_dir(tmp_path, seeded_pyannet(), seeded_wespeaker())
    sd = seeded_wespeaker().state_dict()
    sd.pop("resnet.layer3.2.conv1.weight")
    pm.save_checkpoint(str(tmp_path / "embedding" / "pytorch_model.bin"), sd, WESPEAKER_HPARAMS,
                       pm.WeSpeakerResNet34.ARCHITECTURE, pm.embedding_specifications())
    with pytest.raises(RuntimeError, match="resnet.layer3.2.conv1.weight"):
        load_tool().main([str(tmp_path), os.path.join(ROOT, "tests", "golden", "sample.wav"), "--oracle-only"])


def test_oracle_half_with_the_vbx_configuration(tmp_path, capsys):
    """the 4.x / community-1 layout: `clustering: VBxClustering` + `plda: $model/plda` -- the tool binds the oracle's VBx
    step (pinned to the reference's VBxClustering by tests/test_reference_pipeline.py) to the directory's PLDA"""
    from conftest import write_pipeline_dir
    from oracle.models import seeded_pyannet, seeded_wespeaker
    from oracle.vbx import synth_plda
    write_pipeline_dir(tmp_path, seeded_pyannet(), seeded_wespeaker(), config_extra={
        "pipeline": {"name": "pyannote.audio.pipelines.SpeakerDiarization",
                     "params": {"clustering": "VBxClustering", "embedding": "$model/embedding",
                                "embedding_batch_size": 32, "embedding_exclude_overlap": True,
                                "plda": "$model/plda", "segmentation": "$model/segmentation",
                                "segmentation_batch_size": 32}},
        "params": {"clustering": {"threshold": 0.6, "Fa": 0.07, "Fb": 0.8},
                   "segmentation": {"min_duration_off": 0.0}}})
    synth_plda(str(tmp_path / "plda"))
    out_json = str(tmp_path / "report.json")
    rc = load_tool().main([str(tmp_path), os.path.join(ROOT, "tests", "golden", "sample.wav"), "--max-seconds", "11",
                           "--chunks", "1", "--oracle-only", "--json", out_json])
    text = capsys.readouterr().out
    rep = json.load(open(out_json))
    assert rc == 0 and rep["clustering"] == "VBxClustering" and "PLDA <-" in text
    assert any(r["stage"] == "pipeline" for r in rep["rows"])
