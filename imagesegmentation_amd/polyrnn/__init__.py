"""Teacher-forced Polygon-RNN models (reference polyrnn/)."""
