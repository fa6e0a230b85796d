"""Configuration surface read by the hot path -- same field names and defaults as the reference's
src/diff_gfdn/config/config.py (FeedbackLoopConfig :43-49, OutputFilterConfig :71-89,
SubbandProcessingConfig :63-68, TrainerConfig :108-182) so YAML files and pickled configs carry
over.  Only fields that the accelerated path reads are interpreted; the rest are carried."""
from enum import Enum
from typing import List, Optional, Tuple

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, model_validator


class CouplingMatrixType(Enum):
    SCALAR = "scalar_matrix"
    FILTER = "filter_matrix"
    RANDOM = "random_matrix"

    def __repr__(self) -> str:
        return str(self.value)


class FeatureEncodingType(Enum):
    SINE = "sinusoidal"
    MESHGRID = "meshgrid"

    def __repr__(self) -> str:
        return str(self.value)


class BeamformerType(Enum):
    MAX_DI = "max_di"
    MAX_RE = "max_re"
    BUTTER = "butter"


class FeedbackLoopConfig(BaseModel):
    pu_matrix_order: int = 2 ** 5
    coupling_matrix_type: CouplingMatrixType = CouplingMatrixType.SCALAR
    use_zero_coupling: bool = True


class SubbandProcessingConfig(BaseModel):
    centre_frequency: float
    frequency_range: Tuple
    num_fraction_octaves: int = 3
    use_amp_preserving_filterbank: bool = True


class OutputFilterConfig(BaseModel):
    use_svfs: bool = True
    compress_pole_factor: float = 1.0
    num_hidden_layers: int = 3
    num_neurons_per_layer: int = 2 ** 7
    num_fourier_features: int = 10
    encoding_type: FeatureEncodingType = FeatureEncodingType.SINE
    beamformer_type: Optional[BeamformerType] = None
    use_skip_connections: bool = False


class DecayFilterConfig(BaseModel):
    use_absorption_filters: bool = True
    learn_common_decay_times: bool = False
    initialise_with_opt_values: bool = True


class TestSetConfig(BaseModel):
    __test__ = False
    seed: int = 4314
    ratio: float = 0.1


class TrainerConfig(BaseModel):
    batch_size: int = 32
    num_freq_bins: Optional[int] = None
    device: str = 'cpu'
    train_valid_split: float = 0.8
    hold_out_test_set: Optional[TestSetConfig] = None
    grid_resolution_m: Optional[float] = None
    max_epochs: int = 5
    lr: float = 0.01
    io_lr: float = 0.01
    coupling_angle_lr: float = 0.01
    output_filt_ir_len_ms: float = 500
    use_reg_loss: bool = False
    use_erb_edr_loss: bool = False
    use_colorless_loss: bool = False
    use_asym_spectral_loss: bool = False
    edc_loss_weight: float = 1.0
    edr_loss_weight: float = 1.0
    spectral_loss_weight: float = 1.0
    sparsity_loss_weight: float = 1.0
    use_edc_mask: bool = False
    use_frequency_weighting: bool = False
    subband_process_config: Optional[SubbandProcessingConfig] = None
    train_dir: str = "output/cpu/"
    ir_dir: str = "audio/cpu/"
    save_true_irs: bool = False
    alias_attenuation_db: Optional[int] = None
    reduced_pole_radius: float = Field(default=1.0)

    @model_validator(mode='after')
    def _reduced_pole_radius(self):
        # reference config.py:173-182
        if self.alias_attenuation_db is not None and self.num_freq_bins is not None:
            self.reduced_pole_radius = 10 ** (-abs(self.alias_attenuation_db) / self.num_freq_bins / 20)
        return self


class DiffGFDNConfig(BaseModel):
    seed: int = 46434
    room_dataset_path: str = 'resources/Georg_3room_FDTD/srirs.pkl'
    num_groups: int = 3
    ir_path: Optional[str] = None
    sample_rate: float = 32000.0
    trainer_config: TrainerConfig = TrainerConfig()
    delay_range_ms: List[float] = [20.0, 50.0]
    ambi_order: Optional[int] = None
    num_delay_lines: Optional[int] = 12
    feedback_loop_config: FeedbackLoopConfig = FeedbackLoopConfig()
    decay_filter_config: DecayFilterConfig = DecayFilterConfig()
    output_filter_config: OutputFilterConfig = OutputFilterConfig()
    input_filter_config: Optional[OutputFilterConfig] = OutputFilterConfig()
    model_config = ConfigDict(extra="ignore")

    @model_validator(mode="after")
    def _num_delay_lines(self):
        if self.ambi_order is not None:
            self.num_delay_lines = ((self.ambi_order + 1) ** 2) * self.num_groups
        return self

    @property
    def delay_length_samps(self) -> List[int]:
        """Distinct random primes in the delay range plus the next prime above it
        (reference config.py:262-279; re-seeds numpy's global RNG with ``seed`` on every access)."""
        import sympy as sp
        lo, hi = (int(v * 1e-3 * self.sample_rate) for v in self.delay_range_ms)
        primes = np.array(list(sp.primerange(lo, hi)), dtype=np.int32)
        np.random.seed(self.seed)
        rand_primes = primes[np.random.permutation(len(primes))]
        return np.array(np.r_[rand_primes[:self.num_delay_lines - 1], sp.nextprime(hi)],
                        dtype=np.int32).tolist()
