"""Host-side scheduling decisions of trainer.TrainStep that need no GPU."""
from forwardtacotron_amd import data


def test_predictor_stage_goes_first_only_where_its_bptt_grids_fit_beside_the_postnet_gru(monkeypatch):
    """TrainStep._predictors_first restates the admission arithmetic of ft_rnn_persist.hip (per-XCD demand of a backward
    GRU launch = H/16 chunks x ceil(groups / 8) on 32 one-workgroup CUs, budget 0.75): the single-speaker flagship
    (postnet GRU 256 -> 0.5, widest predictor 128 -> 0.25) runs the predictors' backward stage first, the multispeaker
    model (256-wide pitch predictor -> 0.5) does not, at bs = 32 and bs = 64 alike; the environment knob overrides."""
    from forwardtacotron_amd.model import ForwardTacotron
    from forwardtacotron_amd.multi_model import MultiForwardTacotron
    from forwardtacotron_amd.trainer import TrainStep
    monkeypatch.delenv('FT_PRED_STAGE_FIRST', raising=False)
    single = ForwardTacotron(**data.SINGLESPEAKER_MODEL)
    multi = MultiForwardTacotron(**data.MULTISPEAKER_MODEL)
    assert TrainStep._predictors_first(single, 32) is True
    assert TrainStep._predictors_first(single, 64) is True        # 8 groups still take one chunk row per XCD slot
    assert TrainStep._predictors_first(single, 160) is False      # 20 groups: three rounds of slots
    assert TrainStep._predictors_first(multi, 32) is False
    assert TrainStep._predictors_first(multi, 64) is False
    monkeypatch.setenv('FT_PRED_STAGE_FIRST', '0')
    assert TrainStep._predictors_first(single, 32) is False
    monkeypatch.setenv('FT_PRED_STAGE_FIRST', '1')
    assert TrainStep._predictors_first(multi, 64) is True
