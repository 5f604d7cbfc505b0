"""Host side of the sampling entry points' JOB MODE (no GPU): the reference script's list / directory / BalanceCC-json expansion
(sampling_tv2v.py:106-204, sampling_tv2v_ref.py:124-194), chunking, base-model list, the stand-ins for what the reference computes
with networks outside this build (depth side-car, prompt-seeded tokens), the config-5 deal of chunks to ranks."""
import argparse
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _args(*argv, ref=False):
    from scripts.sampling import sampling_tv2v as S
    p = argparse.ArgumentParser()
    S.add_common_args(p)
    if ref:
        p.add_argument("--prior_type", type=str, default="ref")
        p.add_argument("--reference_path", type=str, default="")
        p.add_argument("--reference_root", type=str, default="")
        p.add_argument("--auto_ref_editing", action="store_true")
    return p.parse_args(list(argv))


def _balancecc(tmp_path):
    """A BalanceCC-style annotation file + frame directories standing in for the .mp4 files (no codec offline)."""
    from PIL import Image
    root = tmp_path / "videos"
    items = [{"Video Type": "Animal", "Video Name": "cat_01", "Editing": [{"Target Prompt": "a tiger walking"}, {"Target Prompt": "a cat, oil painting"}]},
             {"Video Type": "Human", "Video Name": "dance_02", "Editing": [{"Target Prompt": "a robot dancing"}]}]
    rs = np.random.RandomState(0)
    for it in items:
        d = root / it["Video Type"] / it["Video Name"]
        d.mkdir(parents=True)
        for i in range(10):
            Image.fromarray(rs.randint(0, 256, (40, 56, 3)).astype(np.uint8)).save(str(d / f"{i:04d}.png"))
    jpath = tmp_path / "balancecc.json"
    jpath.write_text(json.dumps(items))
    return str(jpath), str(root), items


def test_expand_jobs_single_list_directory_and_json(tmp_path):
    from scripts.sampling import sampling_tv2v as S
    assert S.expand_jobs(_args("--prompt", "a dog", "--video_path", "/v/dog")) == (["a dog"], ["/v/dog"], [], [])
    with pytest.raises(AssertionError, match="prompt and video_path must be provided"):
        S.expand_jobs(_args("--prompt", "a dog"))
    pl, vl = tmp_path / "p.txt", tmp_path / "v.txt"
    pl.write_text("a dog \n a bear\n")
    vl.write_text("/v/one\n/v/two \n")
    assert S.expand_jobs(_args("--prompt_listpath", str(pl), "--video_listpath", str(vl)))[:2] == (["a dog", "a bear"], ["/v/one", "/v/two"])
    with pytest.raises(AssertionError, match="video_listpath must be provided"):
        S.expand_jobs(_args("--prompt_listpath", str(pl)))
    vl.write_text("/v/one\n")
    with pytest.raises(AssertionError, match="The number of prompts and video_paths must be the same, and you provided 2 prompts and 1 video_paths"):
        S.expand_jobs(_args("--prompt_listpath", str(pl), "--video_listpath", str(vl)))
    with pytest.raises(AssertionError, match="Only one of prompt_listpath and videos_directory"):
        S.expand_jobs(_args("--prompt_listpath", str(pl), "--videos_directory", str(tmp_path)))
    vd = tmp_path / "vd"
    (vd / "a red car").mkdir(parents=True)
    (vd / "a blue car").mkdir()
    (vd / "notes.txt").write_text("x")
    pr, vp, sp, _ = S.expand_jobs(_args("--videos_directory", str(vd)))
    assert pr == ["a blue car", "a red car"] and vp == [str(vd / "a blue car"), str(vd / "a red car")] and sp == []
    jpath, root, items = _balancecc(tmp_path)
    with pytest.raises(AssertionError, match="videos_root must be provided"):
        S.expand_jobs(_args("--json_path", jpath))
    pr, vp, sp, _ = S.expand_jobs(_args("--json_path", jpath, "--videos_root", root, "--save_path", "out"))
    assert pr == ["a tiger walking", "a cat, oil painting", "a robot dancing"]
    assert vp == [os.path.join(root, "Animal", "cat_01.mp4")] * 2 + [os.path.join(root, "Human", "dance_02.mp4")]
    assert sp == [os.path.join("out", "Animal", "cat_01", "a tiger walking"), os.path.join("out", "Animal", "cat_01", "a cat, oil painting"),
                  os.path.join("out", "Human", "dance_02", "a robot dancing")]
    # the ref script: one reference image per job, finished jobs dropped (sampling_tv2v_ref.py:155-177)
    with pytest.raises(AssertionError, match="reference_root must be provided"):
        S.expand_jobs(_args("--json_path", jpath, "--videos_root", root, ref=True), with_ref=True)
    out = tmp_path / "out"
    (out / "Animal" / "cat_01" / "a tiger walking").mkdir(parents=True)
    pr, vp, sp, rp = S.expand_jobs(_args("--json_path", jpath, "--videos_root", root, "--reference_root", "/refs", "--save_path", str(out), ref=True),
                                   with_ref=True)
    assert pr == ["a cat, oil painting", "a robot dancing"] and rp == ["/refs/output-a cat, oil painting.png", "/refs/output-a robot dancing.png"]
    assert S.expand_jobs(_args("--prompt", "p", "--video_path", "/v", "--reference_path", "/r.png", ref=True), with_ref=True)[3] == ["/r.png"]


def test_basemodel_list_chunks_and_rank_deal(tmp_path):
    from ccedit_amd.parallel import shard_clips
    from scripts.sampling import sampling_tv2v as S
    from scripts.sampling.util import chunk
    assert S.basemodel_list(_args()) == ["default"]
    assert S.basemodel_list(_args("--basemodel_path", "/m/a.safetensors", "--use_default")) == ["default", "/m/a.safetensors"]
    bl = tmp_path / "b.txt"
    bl.write_text("/m/a.ckpt\n/m/b.ckpt\n")
    assert S.basemodel_list(_args("--basemodel_listpath", str(bl))) == ["/m/a.ckpt", "/m/b.ckpt"]
    with pytest.raises(AssertionError, match="Only one of basemodel_path and basemodel_listpath"):
        S.basemodel_list(_args("--basemodel_path", "/m/a", "--basemodel_listpath", str(bl)))
    # num_samples repeats item by item, --batch_size chunks (sampling_tv2v.py:176-183); the last chunk is short
    rep = [p for p in ["a", "b", "c"] for _ in range(3)]
    assert list(chunk(rep, 4)) == [("a", "a", "a", "b"), ("b", "b", "c", "c"), ("c",)]
    # config 5: chunks dealt round-robin, every chunk exactly once
    deals = [shard_clips(11, r, 8) for r in range(8)]
    assert sorted(i for d in deals for i in d) == list(range(11)) and deals[0] == [0, 8] and deals[3] == [3]
    assert S.job_mode(_args("--prompt", "p", "--video_path", "/v")) and not S.job_mode(_args("--video_path", "/v")) and not S.job_mode(_args("--synthetic"))


def test_video_resolution_depth_sidecar_and_prompt_seeded_text(tmp_path):
    from scripts.sampling import sampling_tv2v as S
    from scripts.sampling.util import keyframe_indices, load_video_keyframes
    jpath, root, items = _balancecc(tmp_path)
    mp4 = os.path.join(root, "Animal", "cat_01.mp4")
    assert S.resolve_video(mp4) == mp4[:-4] and S.resolve_video("/nowhere/x.mp4") == "/nowhere/x.mp4"
    with pytest.raises(NotImplementedError, match="mp4 decoding needs"):
        load_video_keyframes("/nowhere/x.mp4", 20, 3, 4, (32, 48))
    args = _args("--original_fps", "9", "--target_fps", "3", "--num_keyframes", "3", "--H", "32", "--W", "48")
    kf = load_video_keyframes(S.resolve_video(mp4), 9, 3, 3, (32, 48)).permute(1, 0, 2, 3)[None]
    assert kf.shape == (1, 3, 3, 32, 48)
    with pytest.raises(NotImplementedError, match="depth annotators"):
        S.depth_frames(args, mp4, kf)
    args.synthetic = True
    lum = S.depth_frames(args, mp4, kf)
    assert lum.shape == (1, 1, 3, 32, 48) and torch.allclose(lum[0, 0], 0.299 * kf[0, 0] + 0.587 * kf[0, 1] + 0.114 * kf[0, 2])
    # side-car raw depth over ALL frames: the keyframe index rule + bicubic resize, like the frames
    raw = torch.arange(10, dtype=torch.float32)[:, None, None].expand(10, 20, 28).contiguous()
    torch.save(raw, mp4[:-4] + ".depth.pt")
    d = S.depth_frames(args, mp4, kf)
    idx = keyframe_indices(10, 9, 3, 3)
    assert d.shape == (1, 1, 3, 32, 48) and torch.allclose(d[0, 0, :, 5, 5], torch.tensor(idx, dtype=torch.float32), atol=1e-4)
    os.makedirs(str(tmp_path / "depths"))
    torch.save(raw + 100.0, str(tmp_path / "depths" / "cat_01.pt"))
    args.depth_root = str(tmp_path / "depths")
    assert float(S.depth_frames(args, mp4, kf)[0, 0, 0, 0, 0]) == pytest.approx(100.0 + idx[0], abs=1e-3)
    # prompts -> text inputs
    a = _args("--synthetic", "--add_prompt", "masterpiece")
    t1, u1 = S.job_text(a, ["a dog", "a bear"], "cpu", 768)
    t2, _ = S.job_text(a, ["a bear", "a dog"], "cpu", 768)
    assert t1.shape == (2, 77) and t1.dtype == torch.int64 and torch.equal(t1[0], t2[1]) and torch.equal(t1[1], t2[0]) and torch.equal(u1[0], u1[1])
    assert S.job_text(a, ["x"], "cpu", 64)[0].shape == (1, 77, 64)
    a.tokenizer_path = "/tok"
    assert S.job_text(a, ["a dog"], "cpu", 768) == (["masterpiece, a dog"], ["ugly, low quality"])
    with pytest.raises(SystemExit):
        S.job_text(_args(), ["a dog"], "cpu", 768)


def test_perform_save_locally_video_npy_and_mp4(tmp_path):
    from scripts.sampling.util import perform_save_locally_video
    x = torch.rand(2, 3, 4, 8, 8)
    paths = perform_save_locally_video(str(tmp_path / "result"), x, 3, "npy", return_savepaths=True, save_grid=False)
    assert [os.path.basename(p) for p in paths] == ["frames-0000.npy", "frames-0001.npy"]
    assert np.allclose(np.load(paths[1]), x[1].permute(1, 2, 3, 0).numpy())
    more = perform_save_locally_video(str(tmp_path / "result"), x[:1], 3, "npy", return_savepaths=True, save_grid=False)
    assert os.path.basename(more[0]) == "frames-0002.npy"            # counts on from what the directory holds (util.py:300-306)
    with pytest.raises(NotImplementedError, match="mp4 encoding needs"):
        perform_save_locally_video(str(tmp_path / "r2"), x, 3, "mp4")
