SPAN_LIB=variants/tNN.so python profiles/span_probe.py 20 100
