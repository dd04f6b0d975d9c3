"""Drop-in counterparts of the reference's model package for the hot path:
`from model.influentialRS import IRSNN, InfluentialNet` (pipeline.py:15),
`from model.uRS import SampleNet`, `from model.evaluator import Evaluator`
(evaluator_pipeline.py:17-18), `from model.layers import PositionalEncoding,
get_item_index` (influentialRS.py:19) keep working with this package in place
of the reference's `model/` directory."""
