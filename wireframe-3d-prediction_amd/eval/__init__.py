"""Drop-in `eval` package of the reference (eval/ap_calculator.py): Building3D wireframe metrics — row f-4."""
