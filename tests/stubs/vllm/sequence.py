class IntermediateTensors(dict):
    pass
