"""The config surface (reference options.py:9-480) is reproduced flag for flag.  Golden =
tests/golden/options_surface.json, dumped from the reference's own argparse by make_golden.py."""
import json
import os

from fusiondepth_amd.options import MonodepthOptions

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "options_surface.json")


def test_every_reference_flag_with_same_default_type_choices():
    want = json.load(open(GOLD))
    parser = MonodepthOptions().parser
    got = {}
    for a in parser._actions:
        if a.option_strings and a.dest != "help":
            got[a.dest] = {"flag": a.option_strings[0], "default": a.default, "nargs": a.nargs,
                           "type": getattr(a.type, "__name__", None), "choices": list(a.choices) if a.choices else None,
                           "action": type(a).__name__}
    assert sorted(got) == sorted(want), (sorted(set(want) - set(got)), sorted(set(got) - set(want)))
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])


def test_store_false_quirks_and_string_booleans():
    o = MonodepthOptions().parse([])
    assert o.beam_encoder is True and o.need_4beam is True and o.need_2_channel is True
    assert o.trainer_siloss_all_scale is True and o.gdc_loss_only_on_scale_0 is True and o.completion_siloss is True
    assert o.trainer_siloss == "true" and o.catxy == "true" and o.refine_depthnet_with_beam == "false"
    assert o.num_layers == 50 and o.batch_size == 5 and o.frame_ids == [0, -1, 1] and o.scales == [0, 1, 2, 3]
    o = MonodepthOptions().parse(["--beam_encoder", "--num_layers", "18", "--batch_size", "12", "--scales", "0", "2"])
    assert o.beam_encoder is False and o.num_layers == 18 and o.scales == [0, 2]
